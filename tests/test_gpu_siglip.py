"""SigLIP attention-pool head on the GPU: the native module (autograd path) and the fused engine
(ep_siglip_head_train_step through the C ABI) against the golden vectors of the real reference and the CPU oracle.
Needs an MI355X (pytest -m gpu).  fp32 tolerances: forward rtol 2e-5 / atol 1e-5; gradients and updated parameters
rtol 2e-4 with an absolute floor of 6e-5 of the tensor's scale."""
import os

import numpy as np
import pytest
import torch

from cases import SIGLIP_CASES, SIGLIP_PARAM_NAMES, SIGLIP_SMALL, STEP_LRS, SiglipCase, make_siglip_inputs, siglip_sub
from oracle import siglip_oracle as SO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# Gradients that are (partly) sums of cancelling terms (see tests/test_siglip_cpu.py): d fc2.bias is exactly zero and
# d proj.bias keeps only its MLP path (BatchNorm removes a uniform shift of the head's output); d kv.bias[:D] is exactly
# zero (a key bias shifts all scores of a head equally) -- the reference holds rounding noise there, the native path
# writes zeros.
NOISE = {"fc2_b": 2e-5, "proj_b": 2e-5, "kv_b": 5e-6}


def load(case):
    return np.load(os.path.join(GOLD, f"siglip_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.siglip import AttentionPoolLatent
    head = torch.nn.Sequential(AttentionPoolLatent(in_features=case.D), probe_heads._batchnorm(case.D),
                               probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(SIGLIP_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def close(name, got, want, rtol=2e-4, floor=6e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", SIGLIP_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_siglip_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    pooled, attn = head[0](x, return_attn=True)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=2e-5,
                               atol=1e-5 * max(1.0, float(np.abs(g["pooled"]).max())))
    np.testing.assert_allclose(attn[:, :, 0].cpu().numpy(), g["attn"], rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=5e-5)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=2e-5)
    keep = (lambda a: a) if case.full else siglip_sub
    for n, p in zip(SIGLIP_PARAM_NAMES, plist):
        gr = p.grad.cpu().numpy()
        close(n, gr if n in SIGLIP_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        if n not in NOISE:
            assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=2e-4, abs=1e-9)
    D = case.D
    assert float(plist[4].grad[:D].abs().max()) == 0.0           # the key-bias gradient is written as exact zeros


@pytest.mark.parametrize("case", SIGLIP_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import SiglipHeadEngine, make_engine
    g, inp = load(case), make_siglip_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, SiglipHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=3e-5)
        for n, p, mu in zip(SIGLIP_PARAM_NAMES, eng.params_list, eng.mu_views()):
            small = n in SIGLIP_SMALL
            pv, mv = p.detach().cpu().numpy(), mu.cpu().numpy()
            close(f"{tag} {n}", pv if small else keep(pv), g[f"{tag}_{n}"], rtol=3e-4, floor=1e-5, abs_floor=NOISE.get(n, 1e-7))
            close(f"{tag} mu {n}", mv if small else keep(mv), g[f"{tag}_mu_{n}"], rtol=1e-3, floor=2e-4,
                  abs_floor=NOISE.get(n, 1e-7))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"], rtol=5e-4, atol=1e-4)


def test_full_size_batch_vs_oracle_indexed_store_and_determinism():
    from efficient_probing_amd.engine import make_engine
    case = SiglipCase("big", B=64, N=256, D=1152, C=1000, seed=3, sharp=True)
    inp = make_siglip_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = SO.make_head(case.D, case.C)
    with torch.no_grad():
        for n, p in zip(SIGLIP_PARAM_NAMES, SO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want).max())))
    t = torch.from_numpy(inp["targets"]).to(DEV)
    store = torch.cat([x, tokens(case, inp["x_buf2"])], dim=0)
    idx = torch.randperm(store.shape[0], device=DEV)[:case.B].to(torch.int32)
    outs = []
    for gathered in (False, True, False):
        eng = make_engine(native_head(case, inp)[0], optimizer="lars")
        if gathered:
            eng.train_step(store[idx.long()].contiguous(), t, lr=0.5)
        else:
            eng.train_step(store, t, lr=0.5, image_index=idx)
        outs.append(eng.flat_p.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
