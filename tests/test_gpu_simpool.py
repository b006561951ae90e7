"""SimPool heads (simpool / esimpool) on the GPU: the native modules (autograd path) and the fused engine
(ep_simpool_head_train_step through the C ABI; per-image-query token passes) against the golden vectors of the real
reference and the CPU oracle.  Needs an MI355X (pytest -m gpu).  fp32 tolerances: forward rtol 2e-5 / atol 1e-5 of the
output scale; gradients and updated parameters rtol 2e-4 with an absolute floor of 5e-5 of the tensor's scale."""
import os

import numpy as np
import pytest
import torch

from cases import (ESIMPOOL_CASES, SIMPOOL_CASES, SIMPOOL_SMALL, STEP_LRS, SimpoolCase, make_simpool_inputs, siglip_sub,
                   simpool_param_names)
from oracle import simpool_oracle as SO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL = SIMPOOL_CASES + ESIMPOOL_CASES
IDS = [f"{c.family}-{c.name}" for c in ALL]


# simpool: d loss / d norm_patches.bias is exactly zero in exact arithmetic (the bias is a constant added to the head's
# output, which BatchNorm removes again): both sides hold rounding noise only
def noise(case):
    return {"norm_b": 2e-6} if case.linears else {}


def load(case):
    return np.load(os.path.join(GOLD, f"{case.family}_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.simpool import SimPool, SimPool_nolinears
    pool = SimPool(dim=case.D) if case.linears else SimPool_nolinears(dim=case.D, num_heads=case.heads)
    head = torch.nn.Sequential(pool, probe_heads._batchnorm(case.D), probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(simpool_param_names(case), plist):
            p.copy_(torch.from_numpy(inp[n]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def close(name, got, want, rtol=2e-4, floor=5e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", ALL, ids=IDS)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_simpool_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    y2, attn = head[0](x, return_attn=True)
    pooled = head[0](x)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=2e-5,
                               atol=1e-5 * max(1.0, float(np.abs(g["pooled"]).max())))
    assert torch.equal(y2, pooled.detach())
    np.testing.assert_allclose(attn[:, :, 0].cpu().numpy(), g["attn"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=1e-4)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=3e-5)
    keep = (lambda a: a) if case.full else siglip_sub
    for n, p in zip(simpool_param_names(case), plist):
        gr = p.grad.cpu().numpy()
        close(n, gr if n in SIMPOOL_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=noise(case).get(n, 1e-7))
        if n not in noise(case):
            assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=3e-4, abs=1e-9)


@pytest.mark.parametrize("case", ALL, ids=IDS)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import SimpoolHeadEngine, make_engine
    g, inp = load(case), make_simpool_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, SimpoolHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        for n, p, mu in zip(simpool_param_names(case), eng.params_list, eng.mu_views()):
            small = n in SIMPOOL_SMALL
            pv, mv = p.detach().cpu().numpy(), mu.cpu().numpy()
            close(f"{tag} {n}", pv if small else keep(pv), g[f"{tag}_{n}"], rtol=3e-4, floor=1e-5, abs_floor=noise(case).get(n, 1e-7))
            close(f"{tag} mu {n}", mv if small else keep(mv), g[f"{tag}_mu_{n}"], rtol=1e-3, floor=2e-4,
                  abs_floor=noise(case).get(n, 1e-7))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"], rtol=5e-4, atol=1e-4)


@pytest.mark.parametrize("linears", [True, False], ids=["simpool", "esimpool"])
def test_full_size_batch_vs_oracle_and_indexed_store_with_cached_tables(linears):
    from efficient_probing_amd import functional as F_
    from efficient_probing_amd.engine import make_engine
    case = SimpoolCase("big", B=64, N=256, D=768, C=100, linears=linears, seed=7, sharp=True)
    inp = make_simpool_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = SO.make_head(case.D, case.C, case.linears)
    with torch.no_grad():
        for n, p in zip(simpool_param_names(case), SO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=4e-5 * max(1.0, float(np.abs(want).max())))
    # a resident store: both tables computed ONCE for the store, batches drawn by index -- equal to gathered batches
    t = torch.from_numpy(inp["targets"]).to(DEV)
    store = torch.cat([x, tokens(case, inp["x_buf2"])], dim=0)
    ts, im = F_.token_stats(store, F_.SIMPOOL_LN_EPS), F_.channel_stats(store)
    idx = torch.randperm(store.shape[0], device=DEV)[:case.B].to(torch.int32)
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e2 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e3 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e1.train_step(store, t, lr=0.5, image_index=idx, token_stats=ts, image_stats=im)
    e2.train_step(store[idx.long()].contiguous(), t, lr=0.5)
    e3.train_step(store, t, lr=0.5, image_index=idx, token_stats=ts)      # mean tokens recomputed for the batch
    assert torch.equal(e1.flat_p, e2.flat_p) and torch.equal(e3.flat_p, e2.flat_p)
    with pytest.raises(RuntimeError, match="token statistics"):
        e3.train_step(store, t, lr=0.5, image_index=idx)


@pytest.mark.parametrize("linears,D", [(True, 768), (False, 768), (True, 1024), (False, 384), (True, 4096)],
                         ids=["simpool-768", "esimpool-768", "simpool-1024", "esimpool-384", "simpool-4096"])
def test_bf16_tokens_and_deterministic_steps(linears, D):
    from efficient_probing_amd.engine import make_engine
    case = SimpoolCase("mid", B=16, N=196, D=D, C=50, linears=linears, seed=5)
    inp = make_simpool_inputs(case)
    x = tokens(case, inp["x_buf"])
    t = torch.from_numpy(inp["targets"]).to(DEV)
    runs = []
    for _ in range(2):
        e = make_engine(native_head(case, inp)[0], optimizer="lars")
        for _ in range(3):
            e.train_step(x, t, lr=0.3)
        runs.append(e.flat_p.clone())
    assert torch.equal(runs[0], runs[1])
    xb = x.to(torch.bfloat16)
    e16 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e32 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e16.train_step(xb, t, lr=0.3)
    e32.train_step(xb.float(), t, lr=0.3)
    np.testing.assert_allclose(e16.flat_p.cpu().numpy(), e32.flat_p.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # against the oracle after one step from the same start
    oh = SO.make_head(case.D, case.C, case.linears)
    with torch.no_grad():
        for n, p in zip(simpool_param_names(case), SO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
    oh.train()
    logits = oh(torch.from_numpy(inp["x_buf"]))
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(inp["targets"]))
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e1.forward_backward(x, t)
    assert e1.read_stats()[0] == pytest.approx(loss.item(), rel=5e-5)
