"""The launch folds of the fused step (round 2) against the unfolded forms of the same arithmetic.

* the contraction dP = dy_q Wv_q with several N-tiles per workgroup (csrc/ep_gemm.hip, `npers`) against fp64, at the
  bench batch -- the only size at which the fold engages (>= 4 tiles per CU);
* the one-call step (engine._train_step_one_call: phases = 3, the optimizer's norms kernel finishes the cls_token
  gradient reduction, csrc/ep_optim.hip) against forward_backward() + optimizer_step() on a twin engine: BIT-equal
  parameters, momentum and gradients -- the deferred stage sums in ep_reduce_partials_kernel's order;
* the softmax-correction rows computed inside the second token pass (PoolParams.dyv, csrc/ep_pool_stream.hip) against
  ep_delta_kernel in front of it (EP_POOL_DELTA=0), and the N-tile walk against one tile per workgroup
  (EP_GEMM_NPERS=1): same losses and parameters to fp32 summation-order noise, in fresh processes.

Needs an MI355X (pytest -m gpu)."""
import os
import subprocess
import sys
import tempfile
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_head(D=768, Q=8, C=1000, seed=0):
    from efficient_probing_amd import probe_heads

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(D, C)
    torch.manual_seed(seed)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
    return enc.head.to(DEV).train()


@pytest.mark.parametrize("B", [1024, 1025])
def test_dP_contraction_with_tile_walk_vs_fp64(B):
    from efficient_probing_amd import functional as F_
    D, Q = 768, 8
    g = torch.Generator(device=DEV).manual_seed(3)
    dy = torch.randn(B, D, device=DEV, generator=g)
    P = torch.randn(B, Q, D, device=DEV, generator=g)
    Wv = torch.randn(D, D, device=DEV, generator=g) * 0.05
    out = F_.project_backward(dy, None, P, Wv, None, True, None, False, False)
    dP = out[0] if isinstance(out, (tuple, list)) else out
    Dq = D // Q
    ref = torch.einsum("bqc,qcd->bqd", dy.double().view(B, Q, Dq), Wv.double().view(Q, Dq, D))
    err = (dP.double() - ref).abs().max().item()
    assert err <= 2e-6 * ref.abs().max().item() + 1e-6, err          # K = 96 fp32 products: ~1e-7 relative


def test_one_call_step_is_bit_equal_to_two_calls():
    from efficient_probing_amd.engine import ProbeHeadEngine
    h1, h2 = make_head(seed=1), make_head(seed=1)
    e1 = ProbeHeadEngine(h1, optimizer="lars", lr=0.3, weight_decay=1e-4)
    e2 = ProbeHeadEngine(h2, optimizer="lars", lr=0.3, weight_decay=1e-4)
    assert e1._one_call_step()
    g = torch.Generator().manual_seed(9)
    for s in range(3):
        # B = 1024: more than 32 partials of the cls_token gradient -> the two-stage reduction whose last stage is deferred
        x = torch.randn(1024, 64, 768, generator=g).to(DEV)
        t = torch.randint(0, 1000, (1024,), generator=g).to(DEV)
        e1.train_step(x, t)                                           # one library call
        e2.forward_backward(x, t); e2.all_reduce_grads(); e2.optimizer_step()
        assert e1.read_stats()[0] == e2.read_stats()[0]
        assert torch.equal(e1.flat_g, e2.flat_g)                      # incl. the gradient the norms kernel wrote back
        assert torch.equal(e1.flat_p, e2.flat_p)
        assert torch.equal(e1.state[0], e2.state[0])
    assert e1.opt_step == e2.opt_step == 3


CODE = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from argparse import Namespace
from efficient_probing_amd import probe_heads
from efficient_probing_amd.engine import ProbeHeadEngine
D = int(os.environ.get("EP_TEST_D", "768")); BF16 = os.environ.get("EP_TEST_BF16", "0") == "1"
class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.head = torch.nn.Linear(D, 1000)
torch.manual_seed(0); enc = Enc()
probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=8, d_out=1, nb_classes=1000))
eng = ProbeHeadEngine(enc.head.to("cuda:0").train(), optimizer="lars", lr=0.4, weight_decay=1e-4)
g = torch.Generator().manual_seed(5)
losses = []
for s in range(3):
    x = torch.randn(1024, 50, D, generator=g).to("cuda:0"); t = torch.randint(0, 1000, (1024,), generator=g).to("cuda:0")
    if BF16: x = x.to(torch.bfloat16)
    eng.train_step(x, t); losses.append(eng.read_stats()[0])
torch.save({"loss": losses, "p": eng.flat_p.cpu()}, sys.argv[1])
'''


@pytest.mark.parametrize("knob", ["EP_POOL_DELTA", "EP_GEMM_NPERS"])
def test_fold_on_and_off_agree(knob):
    off = {"EP_POOL_DELTA": "0", "EP_GEMM_NPERS": "1"}[knob]
    outs = []
    for val in (None, off):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            env = dict(os.environ)
            env.pop(knob, None)
            if val is not None:
                env[knob] = val
            subprocess.run([sys.executable, "-c", CODE, f.name], check=True, env=env, cwd=ROOT)
            outs.append(torch.load(f.name))
    a, b = outs
    assert np.allclose(a["loss"], b["loss"], rtol=2e-6)
    assert torch.allclose(a["p"], b["p"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("D", [192, 640])
def test_delta_fold_with_bf16_tokens_and_narrow_rows(D):
    """bf16-STORED tokens at D % 256 != 0: the in-pass delta item (dy | y in 1-KiB pieces) must fit the ring
    slot of the bf16 instantiation (twice the tokens per tile at half the bytes: the same slot bytes as fp32).  Shapes the
    bf16 matrix-core kernels do not take, so the vector-ALU streaming kernel runs; fold on against ep_delta_kernel."""
    outs = []
    for val in (None, "0"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            env = dict(os.environ)
            env.pop("EP_POOL_DELTA", None)
            env.update({"EP_TEST_D": str(D), "EP_TEST_BF16": "1"})
            if val is not None:
                env["EP_POOL_DELTA"] = val
            subprocess.run([sys.executable, "-c", CODE, f.name], check=True, env=env, cwd=ROOT)
            outs.append(torch.load(f.name))
    a, b = outs
    assert np.allclose(a["loss"], b["loss"], rtol=2e-6)
    assert torch.allclose(a["p"], b["p"], rtol=2e-4, atol=2e-6)


def test_one_call_step_with_gradient_accumulation_is_bit_equal():
    """phases = 3 with accumulate = 1 (second micro-batch of an accum_iter = 2 step, driven through the C ABI struct the
    engine builds): the deferred reduction stage then ADDS to the gradient of the first micro-batch inside the norms kernel."""
    from efficient_probing_amd import functional as F_, _native as N
    from efficient_probing_amd.engine import ProbeHeadEngine
    e1 = ProbeHeadEngine(make_head(seed=2), optimizer="lars", lr=0.3, weight_decay=1e-4, accum_iter=2)
    e2 = ProbeHeadEngine(make_head(seed=2), optimizer="lars", lr=0.3, weight_decay=1e-4, accum_iter=2)
    g = torch.Generator().manual_seed(11)
    xa = torch.randn(1024, 40, 768, generator=g).to(DEV); ta = torch.randint(0, 1000, (1024,), generator=g).to(DEV)
    xb = torch.randn(1024, 40, 768, generator=g).to(DEV); tb = torch.randint(0, 1000, (1024,), generator=g).to(DEV)
    for e in (e1, e2):
        e.forward_backward(xa, ta)                                   # micro-batch 1: gradients only
    # micro-batch 2 on e1: backward (accumulating) + optimizer in ONE call
    xv, bstride = F_.as_token_view(xb)
    ws = e1._workspace(xv.shape[0], xv.shape[1])
    e1.opt_step += 1
    s = e1._step_struct(xv, bstride, tb.to(torch.int64), 3, True, None)
    N.check(e1._call_train(s, ws), "one-call accumulate step")
    e1._micro = 0
    # ... and the engine's own two-call form on e2
    e2.train_step(xb, tb)
    assert e2._micro == 0 and e2.opt_step == e1.opt_step == 1
    assert torch.equal(e1.flat_g, e2.flat_g)
    assert torch.equal(e1.flat_p, e2.flat_p)
    assert torch.equal(e1.state[0], e2.state[0])


@pytest.mark.parametrize("B", [1024, 1025, 2048 + 7])
def test_value_projection_on_32x96_tiles_vs_fp64(B):
    """y_q = P_q Wv_q^T with N = 96 columns per query: the 32 x 96-tile kernel (csrc/ep_gemm.hip: ep_gemm_kk96_kernel),
    which engages from 256 tiles on (B >= 1024 at Q = 8), against fp64; ragged last row tile included."""
    from efficient_probing_amd import functional as F_
    D, Q = 768, 8
    g = torch.Generator(device=DEV).manual_seed(4)
    P = torch.randn(B, Q, D, device=DEV, generator=g)
    Wv = torch.randn(D, D, device=DEV, generator=g) * 0.05
    y = F_.project_forward(P, Wv)
    Dq = D // Q
    ref = torch.einsum("bqd,qcd->bqc", P.double(), Wv.double().view(Q, Dq, D)).reshape(B, D)
    err = (y.double() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item() + 1e-6, err          # K = 768 fp32 products


@pytest.mark.parametrize("B,C", [(1024, 1000), (1024, 1024), (1024, 44)])
def test_dz_on_32x96_tiles_vs_fp64(B, C):
    """dz = dlogits Wc (K = classes: 1000 has a K tail of 8; 44 is shorter than 8 K-tiles and stays on the 64 x 64
    kernel) on the K/T-layout 32 x 96-tile kernel (csrc/ep_gemm.hip: ep_gemm_kt96_kernel) against fp64."""
    from efficient_probing_amd import functional as F_
    D = 768
    g = torch.Generator(device=DEV).manual_seed(8)
    dl = torch.randn(B, C, device=DEV, generator=g)
    z = torch.randn(B, D, device=DEV, generator=g)
    Wc = torch.randn(C, D, device=DEV, generator=g) * 0.05
    out = F_.linear_backward(dl, z, Wc, True, None, None, False)
    dz = out[0] if isinstance(out, (tuple, list)) else out
    ref = dl.double() @ Wc.double()
    err = (dz.double() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item() + 1e-6, err
