"""DOLG spatial-attention head on the GPU: the native module (autograd path) and the fused engine (ep_dolg_head_train_step
through the C ABI; matrix-core contraction + BatchNorm over the token rows + one streaming row kernel per direction) against the golden vectors of the real reference and the
CPU oracle.  Needs an MI355X (pytest -m gpu).  fp32 tolerances: forward rtol 2e-5 / atol 1e-5 of the output scale;
gradients and updated parameters rtol 2e-4 with an absolute floor of 5e-5 of the tensor's scale."""
import os

import numpy as np
import pytest
import torch

from cases import DOLG_CASES, DOLG_PARAM_NAMES, DOLG_SMALL, STEP_LRS, DolgCase, make_dolg_inputs, siglip_sub
from oracle import dolg_oracle as AO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NOISE = {"conv1_b": 2e-6}     # exactly zero in exact arithmetic, see tests/test_dolg_cpu.py


def load(case):
    return np.load(os.path.join(GOLD, f"dolg_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.dolg import SpatialAttention2d
    head = torch.nn.Sequential(SpatialAttention2d(in_c=case.D, s3_dim=case.D), probe_heads._batchnorm(case.D),
                               probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(DOLG_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
        head[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        head[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def close(name, got, want, rtol=2e-4, floor=5e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", DOLG_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_dolg_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    _, attn = head[0](x, return_attn=True)
    attn = attn.reshape(x.shape[0], -1)
    pooled = head[0](x)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=2e-5,
                               atol=1e-5 * max(1.0, float(np.abs(g["pooled"]).max())))
    np.testing.assert_allclose(attn.cpu().numpy(), g["attn"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=1e-4)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=3e-5)
    # the token BatchNorm's buffers moved exactly once (attention() has no side effects)
    np.testing.assert_allclose(head[0].bn.running_mean.cpu().numpy(), g["lars1_tok_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(head[0].bn.running_var.cpu().numpy(), g["lars1_tok_running_var"], rtol=2e-5, atol=1e-6)
    assert int(head[0].bn.num_batches_tracked) == 1
    keep = (lambda a: a) if case.full else siglip_sub
    for n, p in zip(DOLG_PARAM_NAMES, plist):
        gr = p.grad.cpu().numpy()
        close(n, gr if n in DOLG_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        if n not in NOISE:
            assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=3e-4, abs=1e-9)


@pytest.mark.parametrize("case", DOLG_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import DolgHeadEngine, make_engine
    g, inp = load(case), make_dolg_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, DolgHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        for n, p, mu in zip(DOLG_PARAM_NAMES, eng.params_list, eng.mu_views()):
            small = n in DOLG_SMALL
            pv, mv = p.detach().cpu().numpy(), mu.cpu().numpy()
            close(f"{tag} {n}", pv if small else keep(pv), g[f"{tag}_{n}"], rtol=3e-4, floor=1e-5, abs_floor=NOISE.get(n, 1e-7))
            close(f"{tag} mu {n}", mv if small else keep(mv), g[f"{tag}_mu_{n}"], rtol=1e-3, floor=2e-4, abs_floor=NOISE.get(n, 1e-7))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=2e-4, atol=5e-6)
        np.testing.assert_allclose(head[0].bn.running_mean.cpu().numpy(), g[f"{tag}_tok_running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(head[0].bn.running_var.cpu().numpy(), g[f"{tag}_tok_running_var"], rtol=2e-5, atol=1e-6)
        assert int(head[0].bn.num_batches_tracked) == int(g[f"{tag}_tok_nbt"])
    np.testing.assert_allclose(eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"], rtol=5e-4, atol=1e-4)


def test_full_size_batch_vs_oracle_strided_bf16_and_determinism():
    from efficient_probing_amd.engine import make_engine
    case = DolgCase("big", B=32, N=256, D=768, C=100, seed=3, sharp=True)
    inp = make_dolg_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = AO.make_head(case.D, case.C)
    with torch.no_grad():
        for n, p in zip(DOLG_PARAM_NAMES, AO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(want).max())))
    t = torch.from_numpy(inp["targets"]).to(DEV)
    runs = []
    for _ in range(2):
        e = make_engine(native_head(case, inp)[0], optimizer="lars")
        for _ in range(2):
            e.train_step(x, t, lr=0.3)
        runs.append(e.flat_p.clone())
    assert torch.equal(runs[0], runs[1])
    # a strided view and bf16 tokens are compacted / widened to the dense fp32 matrix the contraction needs
    xs = torch.cat([x[:, :1], x], dim=1)[:, 1:]
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars"); e1.train_step(xs, t, lr=0.3)
    e2 = make_engine(native_head(case, inp)[0], optimizer="lars"); e2.train_step(x, t, lr=0.3)
    assert torch.equal(e1.flat_p, e2.flat_p)
    e3 = make_engine(native_head(case, inp)[0], optimizer="lars"); e3.train_step(x.to(torch.bfloat16), t, lr=0.3)
    e4 = make_engine(native_head(case, inp)[0], optimizer="lars"); e4.train_step(x.to(torch.bfloat16).float(), t, lr=0.3)
    assert torch.equal(e3.flat_p, e4.flat_p)
    # a batch of a resident store (image_index): this matrix-core-bound head gathers it into a contiguous tensor -- same bits
    perm = torch.randperm(case.B, device=DEV).to(torch.int32)
    e5 = make_engine(native_head(case, inp)[0], optimizer="lars"); e5.train_step(x, t[perm.long()], lr=0.3, image_index=perm)
    e6 = make_engine(native_head(case, inp)[0], optimizer="lars"); e6.train_step(x[perm.long()].contiguous(), t[perm.long()], lr=0.3)
    assert torch.equal(e5.flat_p, e6.flat_p)
