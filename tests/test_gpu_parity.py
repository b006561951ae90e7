"""Parity of the HIP path (through the C ABI) against the golden vectors of the real reference
and against the numpy oracle.  Needs an MI355X: run with ``pytest -m gpu``.

Tolerances (fp32, north_star "within a stated fp32 tolerance"):
  forward (pooled, attention, logits, loss)      rtol 1e-5 / atol 1e-6 on pooled & attention,
                                                 rtol 1e-4 / atol 1e-5 after BatchNorm (B is tiny
                                                 in the fixtures, BN amplifies rounding by rstd)
  gradients / updated parameters                 rtol 1e-4, atol 2e-5 * max|expected|
"""
import os

import numpy as np
import pytest
import torch
from argparse import Namespace

from cases import (EPCLS_CASES, make_epcls_inputs, CASES, CASE_BY_NAME, STEP_LRS, make_inputs, view_tokens, sub, keeper, assert_mu_close, post_bn_tol,
                   trust_ratio_gaps, assert_amp_bf16_fidelity)
from oracle import ep_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def load(case):
    return np.load(os.path.join(GOLD, f"ep_{case.name}.npz"))


class Enc(torch.nn.Module):
    def __init__(self, dim, classes):
        super().__init__()
        self.head = torch.nn.Linear(dim, classes)


def build_head(case, inp):
    from efficient_probing_amd import probe_heads
    torch.manual_seed(0)
    enc = Enc(case.D, case.C)
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=case.Q, d_out=case.d_out,
                                                nb_classes=case.C))
    head = enc.head
    with torch.no_grad():
        head[0].cls_token.copy_(torch.from_numpy(inp["cls_token"]))
        head[0].v.weight.copy_(torch.from_numpy(inp["v_weight"]))
        head[2].weight.copy_(torch.from_numpy(inp["fc_weight"]))
        head[2].bias.copy_(torch.from_numpy(inp["fc_bias"]))
    return head.to(DEV).train()


def tokens(case, buf):
    xb = torch.from_numpy(buf).to(DEV)
    return xb[:, 1:] if case.strided else xb          # strided: a real non-contiguous view on the GPU


def gtol(want, rtol=1e-4, k=2e-5):
    return dict(rtol=rtol, atol=max(1e-6, k * float(np.abs(want).max())))


def test_native_library_is_loaded():
    from efficient_probing_amd import _native
    lib = _native.load()
    assert lib.ep_version() >= 2
    assert lib.ep_device_cu_count() > 0


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_pool_forward_matches_reference_attention(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_inputs(case)
    x = tokens(case, inp["x_buf"])
    scale = case.D ** -0.5
    P, S, ML = F_.pool_forward(x, torch.from_numpy(inp["cls_token"]).to(DEV), scale)
    A = F_.attention_from_scores(S, ML).cpu().numpy()
    np.testing.assert_allclose(A, g["attn"], rtol=1e-5, atol=1e-6)
    xn = view_tokens(case, inp["x_buf"])
    Pref = np.matmul(g["attn"].astype(np.float64), xn.astype(np.float64))
    np.testing.assert_allclose(P.cpu().numpy(), Pref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(A.sum(-1), 1.0, rtol=0, atol=1e-5)


@pytest.mark.parametrize("tok", ["f32", "bf16"])
@pytest.mark.parametrize("case", EPCLS_CASES, ids=lambda c: c.name)
def test_per_image_queries_module_matches_reference(case, tok):
    """EfficientProbing.forward(x, cls=...) -- per-image queries override the learned ones (reference poolings/ep.py:32-33) --
    through the module: pooled vector, gradient of ``cls`` (one (Q, D) gradient per image) and of v.weight against the
    real reference's (tests/golden/epcls_*.npz); the learned cls_token takes no gradient.  bf16-stored tokens: the same
    checks against the oracle fed the rounded tokens."""
    from efficient_probing_amd.poolings.ep import EfficientProbing
    g = np.load(os.path.join(GOLD, f"epcls_{case.name}.npz"))
    inp = make_epcls_inputs(case)
    pool = EfficientProbing(case.D, num_queries=case.Q, d_out=case.d_out).to(DEV)
    with torch.no_grad():
        pool.cls_token.copy_(torch.from_numpy(inp["cls_token"]))
        pool.v.weight.copy_(torch.from_numpy(inp["v_weight"]))
    x = tokens(case, inp["x_buf"])
    want = dict(pooled=g["pooled"], cls=g["grad_cls"], v=g["grad_v_weight"])
    if tok == "bf16":
        if case.strided:
            pytest.skip("bf16 token views need 16-byte aligned rows: covered by the dense cases")
        x = x.to(torch.bfloat16)
        xr = x.float().cpu().numpy()
        out, cache = O.ep_forward(xr, inp["cls_token"], inp["v_weight"], case.Q, d_out=case.d_out, cls=inp["cls"])
        dy = torch.from_numpy(inp["dy"]).to(torch.bfloat16).float().numpy()     # the module returns x.dtype: the upstream gradient arrives as bf16
        dc, dw = O.ep_backward(dy, cache, inp["v_weight"])
        want = dict(pooled=out, cls=dc, v=keeper(case)(dw))
    cls = torch.from_numpy(inp["cls"]).to(DEV).requires_grad_(True)
    pooled = pool(x, cls=cls)
    assert pooled.dtype == x.dtype
    pooled.float().backward(torch.from_numpy(inp["dy"]).to(DEV)) if tok == "f32" else pooled.backward(torch.from_numpy(inp["dy"]).to(DEV).to(pooled.dtype))
    assert pool.cls_token.grad is None
    ftol = dict(rtol=1e-5, atol=2e-6) if tok == "f32" else dict(rtol=1e-2, atol=1e-2)      # bf16 OUTPUT rounding (module returns x.dtype)
    np.testing.assert_allclose(pooled.detach().float().cpu().numpy(), want["pooled"], **ftol)
    np.testing.assert_allclose(cls.grad.cpu().numpy(), want["cls"], **gtol(want["cls"]))
    np.testing.assert_allclose(keeper(case)(pool.v.weight.grad.cpu().numpy()), want["v"], **gtol(want["v"]))
    with pytest.raises(ValueError):
        pool(x, cls=cls[:, :-1])


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_streaming_and_generic_kernels_agree(case):
    from efficient_probing_amd import functional as F_, _native
    inp = make_inputs(case)
    x = tokens(case, inp["x_buf"])
    cls = torch.from_numpy(inp["cls_token"]).to(DEV)
    scale = case.D ** -0.5
    lib = _native.load()
    rng = np.random.default_rng(5)
    dP = torch.from_numpy(rng.standard_normal((case.B, case.Q, case.D), dtype=np.float32)).to(DEV)
    outs = []
    for mode in (1, 0, 2, 3):    # generic (reference of this test), automatic, vector-ALU streaming, all-matrix-core
        lib.ep_debug_force_generic_pool(mode)
        try:
            P, S, ML = F_.pool_forward(x, cls, scale)
            ML2 = ML.clone()
            ML2[:, :, 2] = 0.25
            dcls = F_.pool_backward(x, S, ML2, dP, scale)
            torch.cuda.synchronize()
            outs.append([t.cpu().numpy() for t in (P, S, dcls)] + [F_.attention_from_scores(S, ML).cpu().numpy()])
        finally:
            lib.ep_debug_force_generic_pool(0)
    for which, other in (("auto", outs[1]), ("valu-stream", outs[2]), ("all-mfma", outs[3])):
        for a, b, name in zip(other, outs[0], ("P", "S", "dcls", "A")):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5 * max(1e-3, float(np.abs(b).max())),
                                       err_msg=f"{which} {name}")


# The query-chunk dispatch (csrc/ep_pool.hip: query_chunk) is chosen per process from EP_POOL_QCHUNK: 1 = the default (f32 tokens at
# D <= 768 and Q > 16: forward as 16-query chunks on the all-matrix-core kernel, backward ONE pass on the vector-ALU kernel --
# two kernel families sharing S / ML), 2 = both directions chunked, 3 = neither.  Every setting in a fresh process against the
# generic kernel on the two Q = 32 reference cases, and the default against the reference's golden pooled vector / attention.
_QCHUNK_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["EP_TEST_ROOT"], "tests", "golden")); sys.path.insert(0, os.environ["EP_TEST_ROOT"])
from cases import CASE_BY_NAME, make_inputs
from efficient_probing_amd import functional as F_, _native
lib = _native.load()
for name in ("vitb14_q32", "vitl_q32_dout2"):
    case = CASE_BY_NAME[name]
    inp = make_inputs(case)
    x = torch.from_numpy(inp["x_buf"]).cuda()
    for storage in ("f32", "bf16"):
        xs = x.to(torch.bfloat16) if storage == "bf16" else x
        cls = torch.from_numpy(inp["cls_token"]).cuda()[0] * (40.0 if storage == "f32" else 25.0)     # a sharp softmax
        dP = torch.from_numpy(np.random.default_rng(5).standard_normal((case.B, case.Q, case.D), dtype=np.float32)).cuda()
        outs = []
        for mode in (1, 0):
            lib.ep_debug_force_generic_pool(mode)
            P, S, ML = F_.pool_forward(xs, cls, case.D ** -0.5)
            ML2 = ML.clone(); ML2[:, :, 2] = 0.25
            dcls = F_.pool_backward(xs, S, ML2, dP, case.D ** -0.5)
            torch.cuda.synchronize()
            outs.append([t.cpu().numpy() for t in (P, S, dcls)])
        lib.ep_debug_force_generic_pool(0)
        fams = [lib.ep_pool_kernel_name_ex(case.B, case.N, case.D, case.Q, b, 1 if storage == "bf16" else 0).decode() for b in (0, 1)]
        for a, b, n in zip(outs[1], outs[0], ("P", "S", "dcls")):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5 * max(1e-3, float(np.abs(b).max())), err_msg=f"{name} {storage} {n} {fams}")
        print("FAMILY", name, storage, fams[0], fams[1])
print("QCHUNK_OK")
"""


@pytest.mark.parametrize("qchunk", ["1", "2", "3"])
def test_query_chunk_dispatch_agrees_with_generic_in_fresh_process(qchunk):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EP_POOL_QCHUNK=qchunk, EP_TEST_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _QCHUNK_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "QCHUNK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    fam = {tuple(l.split()[1:3]): l.split()[3:5] for l in r.stdout.splitlines() if l.startswith("FAMILY")}
    assert all("generic" not in k for v in fam.values() for k in v), fam


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_module_forward_backward_golden(case):
    """Drop-in path: Sequential(EfficientProbing, BatchNorm1d, Linear) under autograd."""
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_inputs(case)
    head = build_head(case, inp)
    x = tokens(case, inp["x_buf"])
    t = torch.from_numpy(inp["targets"]).to(DEV)
    pooled = head[0](x)
    z = head[1](pooled)
    logits = head[2](z)
    loss, stats = F_.cross_entropy_loss(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["z"], **post_bn_tol(case))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], **post_bn_tol(case))
    assert float(loss) == pytest.approx(float(g["loss"]), rel=2e-5)
    # ... and the distance of this fp32 head from the reference head under the published --amp bfloat16 protocol
    assert_amp_bf16_fidelity(logits.detach().cpu().numpy(), float(loss), g, err_msg=case.name)
    st = stats.cpu().numpy()
    assert st[1] * 100.0 / case.B == pytest.approx(float(g["acc1"]))
    assert st[2] * 100.0 / case.B == pytest.approx(float(g["acc5"]))
    assert st[3] == 0
    keep = keeper(case)
    got = {"cls_token": head[0].cls_token.grad, "v_weight": head[0].v.weight.grad,
           "fc_weight": head[2].weight.grad, "fc_bias": head[2].bias.grad}
    for n, gt in got.items():
        a = gt.detach().cpu().numpy()
        a = a if n in ("cls_token", "fc_bias") else keep(a)
        np.testing.assert_allclose(a, g[f"grad_{n}"], **gtol(g[f"grad_{n}"]), err_msg=n)
    # torch's own loss on our logits gives the same gradient path through the native modules
    assert int(head[1].num_batches_tracked) == 1


@pytest.mark.parametrize("opt", ["lars", "sgd"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_fused_engine_steps_golden(case, opt):
    from efficient_probing_amd.engine import ProbeHeadEngine
    g, inp = load(case), make_inputs(case)
    if f"{opt}1_loss" not in g:
        pytest.skip("not recorded")
    head = build_head(case, inp)
    eng = ProbeHeadEngine(head, optimizer=opt, weight_decay=case.weight_decay)
    keep = keeper(case)
    names = ["cls_token", "v_weight", "fc_weight", "fc_bias"]
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        tg = inp["targets"] if step % 2 == 0 else inp["targets2"]
        eng.train_step(tokens(case, xb), torch.from_numpy(tg).to(DEV), lr=STEP_LRS[step % len(STEP_LRS)])
        loss, top1, top5, bad = eng.read_stats()
        tag = f"{opt}{step + 1}"
        assert loss == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        assert bad == 0 and int(eng.found_inf.item()) == 0
        for n, p in zip(names, eng.params_list):
            a = p.detach().cpu().numpy()
            a = a if n in ("cls_token", "fc_bias") else keep(a)
            # later steps at lr 1.6 / 0.8 with B = 3..4 amplify fp32 rounding through BN's 1/sigma
            np.testing.assert_allclose(a, g[f"{tag}_{n}"], rtol=1e-4, atol=3e-6 if step == 0 or case.B >= 64 else 3e-5,
                                       err_msg=f"{tag} {n}")
        if opt == "lars":
            for n, mu in zip(names, eng.mu_views()):
                a = mu.detach().cpu().numpy()
                a = a if n in ("cls_token", "fc_bias") else keep(a)
                want = g[f"{tag}_mu_{n}"]
                # the golden trust ratio carries torch-CPU's fp32 norm error as one common factor per step; the fixture
                # records it (q64 / q32), so the factor is predicted, not fitted (cases.assert_mu_close)
                assert_mu_close(a, want, err_msg=f"mu {n}", gaps=trust_ratio_gaps(g, n, step + 1))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=1e-5, atol=1e-6)
        assert int(head[1].num_batches_tracked) == int(g[f"{tag}_nbt"])
    if opt == "lars":
        ev = eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy()
        np.testing.assert_allclose(ev, g["eval_logits"], **(dict(rtol=2e-5, atol=2e-5) if case.B >= 64 else dict(rtol=2e-4, atol=2e-4)))
        head.eval()
        with torch.no_grad():
            ev2 = head(tokens(case, inp["x_buf"])).cpu().numpy()
        np.testing.assert_allclose(ev2, ev, rtol=1e-6, atol=1e-6)
        if "eval_logits_fp16_autocast" in g.files:
            # the reference's own evaluation mode (fp16 autocast, engine_finetune.py:131) through the precision switch:
            # an fp16 result, pinned to a few fp16 ulps of the logits' scale; the fp32 evaluation above sits further away
            want = g["eval_logits_fp16_autocast"]
            ev16 = eng.eval_logits(tokens(case, inp["x_buf"]), precision="fp16_autocast").cpu().numpy()
            ulp = 2.0 ** -11 * float(np.abs(want).max())
            np.testing.assert_allclose(ev16, want, rtol=0, atol=6 * ulp)
            assert (ev16.argmax(1) != want.argmax(1)).sum() <= max(1, case.B // 32)      # near-ties at an untrained head


ADHOC = [dict(B=16, N=50, D=256, Q=8, C=33), dict(B=6, N=37, D=2048, Q=8, C=20), dict(B=5, N=20, D=4096, Q=4, C=12),
         dict(B=132, N=9, D=2048, Q=8, C=37)]    # D >= 2048 and B >= 128: all six contractions on the bf16-plane kernel


@pytest.mark.parametrize("shape", ADHOC, ids=["d256", "wide2048", "wide4096", "wide2048_b132"])
def test_engine_matches_oracle_multi_step_random(shape):
    """Oracle (not golden) parity on shapes no fixture has (incl. the wide-row kernels), 4 LARS steps with weight
    decay."""
    from efficient_probing_amd.engine import ProbeHeadEngine
    from cases import Case
    case = Case("adhoc", seed=11, weight_decay=1e-3, **shape)
    inp = make_inputs(case)
    head = build_head(case, inp)
    eng = ProbeHeadEngine(head, optimizer="lars", weight_decay=case.weight_decay)
    st = O.HeadState(cls_token=inp["cls_token"].copy(), v_weight=inp["v_weight"].copy(),
                     fc_weight=inp["fc_weight"].copy(), fc_bias=inp["fc_bias"].copy(),
                     running_mean=np.zeros(case.D, np.float32), running_var=np.ones(case.D, np.float32),
                     num_queries=case.Q, d_out=1)
    for step in range(4):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        tg = inp["targets"] if step % 2 == 0 else inp["targets2"]
        ref = O.head_train_step(st, xb, tg, lr=0.3, weight_decay=case.weight_decay)
        eng.train_step(torch.from_numpy(xb).to(DEV), torch.from_numpy(tg).to(DEV), lr=0.3)
        loss = eng.read_stats()[0]
        assert loss == pytest.approx(float(ref["loss"]), rel=5e-5)
    for n, p in zip(O.PARAM_ORDER, eng.params_list):
        np.testing.assert_allclose(p.detach().cpu().numpy(), getattr(st, n), rtol=2e-4, atol=1e-5, err_msg=n)


def test_lars_optimizer_class_edge_cases():
    """util/lars.py:26-29 edge cases through the drop-in optimizer class."""
    from efficient_probing_amd.util.lars import LARS
    g = np.load(os.path.join(GOLD, "lars_edges.npz"))
    ps = [torch.nn.Parameter(torch.from_numpy(g[f"p{i}_before"].copy()).to(DEV)) for i in range(5)]
    opt = LARS(ps, lr=0.5, weight_decay=0.0)
    for step in (1, 2):
        for i, p in enumerate(ps):
            p.grad = torch.from_numpy(g[f"g{i}"].copy()).to(DEV)
        opt.step()
        for i, p in enumerate(ps):
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"p{i}_after{step}"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(opt.state[p]["mu"].cpu().numpy(), g[f"mu{i}_after{step}"], rtol=1e-5, atol=1e-7)
    assert int(opt.last_found_inf.item()) == 0
    ps2 = [torch.nn.Parameter(torch.from_numpy(g[f"p{i}_before"].copy()).to(DEV)) for i in range(5)]
    opt2 = LARS(ps2, lr=0.5, weight_decay=0.01)
    for i, p in enumerate(ps2):
        p.grad = torch.from_numpy(g[f"g{i}"].copy()).to(DEV)
    opt2.step()
    for i, p in enumerate(ps2):
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"wd_p{i}_after1"], rtol=1e-5, atol=1e-7)
    # overflow: nothing moves, flag raised (GradScaler contract)
    before = [p.detach().clone() for p in ps2]
    ps2[0].grad = torch.full_like(ps2[0], float("inf"))
    opt2.step(inv_scale=1.0 / 65536)
    assert int(opt2.last_found_inf.item()) == 1
    for p, b in zip(ps2, before):
        assert torch.equal(p.detach(), b)


def test_bad_arguments_fail_loudly():
    from efficient_probing_amd import functional as F_
    x = torch.randn(2, 5, 64, device=DEV)
    with pytest.raises(RuntimeError):
        F_.pool_forward(torch.randn(2, 5, 64), torch.randn(4, 64, device=DEV), 0.125)      # CPU tokens
    with pytest.raises(RuntimeError, match="multiple of 4"):
        F_.pool_forward(torch.randn(2, 5, 66, device=DEV), torch.randn(4, 66, device=DEV), 0.125)
    from efficient_probing_amd.poolings.ep import EfficientProbing
    with pytest.raises(ValueError):
        EfficientProbing(dim=64, num_queries=5)


# ------------------------------------------------------------------------------------------
# full-size property tests (BASELINE.json config 2 and the north-star shape): the oracle is too slow
# there, so parity is established through size-independent properties.
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(192, 256, 768, 8), (160, 197, 768, 8), (64, 196, 1024, 8), (48, 256, 1152, 8),
                                   (24, 196, 4096, 8), (20, 197, 2048, 5), (300, 31, 4096, 8)],
                         ids=["vitb14", "vitb16", "vitl16", "so400m", "vit7b", "wide2048_odd", "vit7b_many_images"])
def test_full_size_properties(shape):
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    if D >= 2048:
        assert _native.load().ep_pool_kernel_name(B, Nn, D, Q, 0).decode() == "ep_pool_wide_fwd_kernel"
    gen = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(B, Nn, D, generator=gen).to(DEV)
    cls = (torch.randn(Q, D, generator=gen) * 0.5).to(DEV)
    scale = D ** -0.5
    P, S, ML = F_.pool_forward(x, cls, scale)
    A = F_.attention_from_scores(S, ML)
    # (1) softmax rows sum to one, (2) P equals A @ x computed by an independent fp64 matmul
    assert torch.allclose(A.sum(-1), torch.ones_like(A.sum(-1)), atol=2e-5)
    Pref = torch.matmul(A.double(), x.double())
    assert torch.allclose(P.double(), Pref, rtol=1e-5, atol=2e-6)
    # (3) scores are the scaled dot products
    Sref = torch.matmul((cls * scale).double(), x.double().transpose(1, 2))
    assert torch.allclose(S.double(), Sref, rtol=1e-5, atol=1e-5)
    # (4) token-permutation invariance of the pooled vectors
    perm = torch.randperm(Nn, generator=gen).to(DEV)
    P2, _, _ = F_.pool_forward(x[:, perm].contiguous(), cls, scale)
    assert torch.allclose(P, P2, rtol=1e-5, atol=2e-6)
    # (5) batch independence: an image pooled alone gives the same row
    P3, _, _ = F_.pool_forward(x[B // 2:B // 2 + 1], cls, scale)
    assert torch.allclose(P[B // 2:B // 2 + 1], P3, rtol=0, atol=1e-6)
    # (6) backward is linear in dP and matches an fp64 evaluation of the formula
    dP = torch.randn(B, Q, D, generator=gen).to(DEV)
    ML2 = ML.clone()
    delta = (dP.double() * P.double()).sum(-1)
    ML2[:, :, 2] = delta.float()
    dcls = F_.pool_backward(x, S, ML2, dP, scale)
    dA = torch.matmul(dP.double(), x.double().transpose(1, 2))
    dS = A.double() * (dA - delta[..., None])
    ref = scale * torch.matmul(dS, x.double()).sum(0)
    assert torch.allclose(dcls.double(), ref, rtol=1e-4, atol=2e-5 * float(ref.abs().max()))
    dcls2 = F_.pool_backward(x, S, ML2, 2 * dP, scale)
    ML3 = ML2.clone(); ML3[:, :, 2] *= 2
    dcls2 = F_.pool_backward(x, S, ML3, 2 * dP, scale)
    assert torch.allclose(dcls2, 2 * dcls, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(7, 33, 2048, 8), (9, 40, 4096, 8), (5, 17, 4096, 3)], ids=["d2048", "d4096", "d4096_q3"])
def test_wide_row_and_generic_kernels_agree(shape):
    """The wide-row kernels (row split across the waves) against the generic kernel on the same inputs, including
    an indexed (resident-store) batch."""
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    gen = torch.Generator(device="cpu").manual_seed(8)
    store = torch.randn(B + 3, Nn, D, generator=gen).to(DEV)
    idx = torch.randperm(B + 3, generator=gen)[:B].to(torch.int32).to(DEV)
    x = store[idx.long()].contiguous()
    cls = (torch.randn(Q, D, generator=gen) * 0.3).to(DEV)
    dP = torch.randn(B, Q, D, generator=gen).to(DEV)
    scale = D ** -0.5
    lib = _native.load()
    outs = []
    for mode in (1, 0):
        lib.ep_debug_force_generic_pool(mode)
        try:
            P, S, ML = F_.pool_forward(x, cls, scale)
            ML2 = ML.clone(); ML2[:, :, 2] = 0.25
            dcls = F_.pool_backward(x, S, ML2, dP, scale)
            lse = ML[:, :, 0] + torch.log(ML[:, :, 1])        # (running max, sum) is representation specific; their
            outs.append([t.cpu().numpy() for t in (P, S, dcls, lse)])   # log-sum-exp is not
        finally:
            lib.ep_debug_force_generic_pool(0)
    for a, b, name in zip(outs[1], outs[0], ("P", "S", "dcls", "logsumexp")):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5 * max(1e-3, float(np.abs(b).max())), err_msg=name)
    Pi, Si, MLi = F_.pool_forward(store, cls, scale, image_index=idx)
    assert torch.equal(Pi.cpu(), torch.from_numpy(outs[1][0])) and torch.equal(Si.cpu(), torch.from_numpy(outs[1][1]))


@pytest.mark.parametrize("shape", [(1, 1, 64, 1), (2, 3, 64, 4), (1, 5, 768, 8), (3, 1, 4096, 8), (1, 2, 2048, 8), (2, 7, 1152, 8),
                                   (1, 9, 1024, 16), (5, 4, 200, 5)], ids=lambda s: "x".join(map(str, s)))
def test_degenerate_shapes(shape):
    """One image, one token, fewer tokens than a ring tile, a single query: every kernel family against the numpy
    oracle (forward) and against the generic kernel (backward)."""
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    rng = np.random.default_rng(9)
    x = rng.standard_normal((B, Nn, D), dtype=np.float32)
    cls = (0.3 * rng.standard_normal((1, Q, D), dtype=np.float32)).astype(np.float32)
    scale = D ** -0.5
    xd, cd = torch.from_numpy(x).to(DEV), torch.from_numpy(cls).to(DEV)
    P, S, ML = F_.pool_forward(xd, cd, scale)
    A = F_.attention_from_scores(S, ML).cpu().numpy()
    want_A = O.ep_attention(x, cls)
    np.testing.assert_allclose(A, want_A, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(P.cpu().numpy(), np.einsum("bqn,bnd->bqd", want_A, x), rtol=2e-5, atol=2e-6)
    dP = torch.from_numpy(rng.standard_normal((B, Q, D), dtype=np.float32)).to(DEV)
    ML2 = ML.clone(); ML2[:, :, 2] = (dP * P).sum(-1)
    got = F_.pool_backward(xd, S, ML2, dP, scale).cpu().numpy()
    lib = _native.load()
    lib.ep_debug_force_generic_pool(1)
    try:
        ref = F_.pool_backward(xd, S, ML2, dP, scale).cpu().numpy()
    finally:
        lib.ep_debug_force_generic_pool(0)
    if Nn == 1:        # one token: A = 1, so dS = dA - dP.P = 0 exactly; both sides hold rounding noise of the two terms
        assert float(np.abs(got).max()) < 2e-5 and float(np.abs(ref).max()) < 2e-5
    else:
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * max(1e-3, float(np.abs(ref).max())))


def test_empty_batch_and_bad_token_arguments_fail_loudly():
    from efficient_probing_amd import functional as F_
    cls = torch.zeros(1, 4, 64, device=DEV)
    with pytest.raises(RuntimeError):
        F_.pool_forward(torch.zeros(0, 5, 64, device=DEV), cls, 0.125)          # B = 0
    with pytest.raises(RuntimeError):
        F_.pool_forward(torch.zeros(2, 5, 62, device=DEV), torch.zeros(1, 4, 62, device=DEV), 0.125)   # D % 4 != 0
    with pytest.raises(ValueError):
        F_.pool_forward(torch.zeros(2, 64, device=DEV), cls, 0.125)             # not (B, N, D)


@pytest.mark.parametrize("shape", [(6, 200, 768, 8), (5, 197, 384, 1), (4, 196, 1024, 8), (4, 256, 1152, 8), (3, 100, 4096, 8),
                                   (3, 64, 2048, 4), (4, 90, 256, 16)], ids=lambda s: "x".join(map(str, s)))
def test_online_softmax_under_adversarial_score_order(shape):
    """Scores that keep growing along the token axis (every tile raises the running maximum, so the lazy-max rescale
    path runs all the time) and span +-60 (exp would overflow without the running maximum): every kernel family must
    match an fp64 softmax, forward and backward, with no non-finite value."""
    from efficient_probing_amd import functional as F_
    B, Nn, D, Q = shape
    gen = torch.Generator(device="cpu").manual_seed(13)
    cls = torch.randn(Q, D, generator=gen)
    cls = cls / cls.norm(dim=1, keepdim=True)
    x = torch.randn(B, Nn, D, generator=gen) * 0.05
    ramp = torch.linspace(-60.0, 60.0, Nn)                                    # target score of token n for query 0
    x = x + ramp[None, :, None] * cls[0][None, None, :]                        # along cls[0]: score_0(n) ~ ramp[n]
    xd, cd = x.to(DEV), cls.to(DEV)
    P, S, ML = F_.pool_forward(xd, cd, 1.0)
    assert torch.isfinite(P).all() and torch.isfinite(S).all() and torch.isfinite(ML[:, :, :2]).all()
    Sref = torch.matmul(cls.double(), x.double().transpose(1, 2))
    Aref = torch.softmax(Sref, dim=-1)
    Pref = torch.matmul(Aref, x.double())
    np.testing.assert_allclose(S.cpu().double().numpy(), Sref.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(P.cpu().double().numpy(), Pref.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(F_.attention_from_scores(S, ML).cpu().double().numpy(), Aref.numpy(), rtol=2e-4, atol=1e-7)
    dP = torch.randn(B, Q, D, generator=gen)
    delta = (dP.double() * Pref).sum(-1)
    ML2 = ML.clone(); ML2[:, :, 2] = delta.float().to(DEV)
    dcls = F_.pool_backward(xd, S, ML2, dP.to(DEV), 1.0)
    dA = torch.matmul(dP.double(), x.double().transpose(1, 2))
    ref = torch.matmul(Aref * (dA - delta[..., None]), x.double()).sum(0)
    assert torch.isfinite(dcls).all()
    np.testing.assert_allclose(dcls.cpu().double().numpy(), ref.numpy(), rtol=2e-4, atol=5e-5 * float(ref.abs().max()))


def test_label_outside_the_class_range_is_flagged_not_read():
    """ep_ce_kernel read row[target] unchecked.  A label of -1 (an ignore_index) or >= C now marks the row as
    bad (stats[3], which stops the training loop), and that row contributes no loss and no gradient."""
    from efficient_probing_amd import functional as F_
    torch.manual_seed(3)
    logits = torch.randn(6, 10, device=DEV, requires_grad=True)
    good = torch.tensor([1, 2, 3, 4, 5, 6], device=DEV)
    for badval in (-1, 10, 1 << 40):
        t = good.clone(); t[2] = badval
        lg = logits.detach().clone().requires_grad_(True)
        loss, stats = F_.cross_entropy_loss(lg, t)
        loss.backward()
        st = stats.cpu().numpy()
        assert st[3] == 1                                        # one flagged row
        assert torch.isfinite(loss) and torch.isfinite(lg.grad).all()
        assert float(lg.grad[2].abs().max()) == 0.0              # no gradient from the flagged row
        keep = [0, 1, 3, 4, 5]
        ref = torch.nn.functional.cross_entropy(logits.detach()[keep], good[keep], reduction="sum") / 6
        assert float(loss) == pytest.approx(float(ref), rel=1e-5)
