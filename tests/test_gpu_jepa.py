"""V-JEPA attentive pooler on the GPU: the native module (autograd path) and the fused engine (ep_jepa_head_train_step
through the C ABI; LayerNorm-of-tokens mode of the token passes) against the golden vectors of the real reference and the
CPU oracle.  Needs an MI355X (pytest -m gpu).  fp32 tolerances as for the other heads: forward rtol 2e-5 / atol 1e-5 of the
output scale; gradients and updated parameters rtol 2e-4 with an absolute floor of 3e-5 of the tensor's scale."""
import os

import numpy as np
import pytest
import torch

from cases import JEPA_CASES, JEPA_PARAM_NAMES, JEPA_SMALL, STEP_LRS, JepaCase, make_jepa_inputs, siglip_sub
from oracle import jepa_oracle as JO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NOISE = {"kv_b": 5e-6, "fc2_b": 1e-4, "proj_b": 1e-4, "query": 1e-4, "n1_b": 1e-5}     # see tests/test_jepa_cpu.py


def load(case):
    return np.load(os.path.join(GOLD, f"jepa_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.jepa import AttentivePooler
    head = torch.nn.Sequential(AttentivePooler(embed_dim=case.D, num_heads=case.heads), probe_heads._batchnorm(case.D),
                               probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(JEPA_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def close(name, got, want, rtol=2e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", JEPA_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_jepa_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    pooled = head[0](x)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=2e-5,
                               atol=1e-5 * max(1.0, float(np.abs(g["pooled"]).max())))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=1e-4)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=3e-5)
    keep = (lambda a: a) if case.full else siglip_sub
    D = case.D
    for n, p in zip(JEPA_PARAM_NAMES, plist):
        gr = p.grad.cpu().numpy()
        # floor 5e-4 of the tensor's scale: with six images BatchNorm's backward amplifies the upstream gradients, and
        # every entry is the end of a chain of D- and 4D-long fp32 contractions in a different summation order
        close(n, gr if n in JEPA_SMALL else keep(gr), g[f"grad_{n}"], floor=5e-4, abs_floor=NOISE.get(n, 1e-7))
        if n not in NOISE:
            assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=3e-4, abs=1e-9)
    assert float(plist[6].grad[:D].abs().max()) == 0.0               # key half of kv.bias: exact zeros


@pytest.mark.parametrize("case", JEPA_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import JepaHeadEngine, make_engine
    g, inp = load(case), make_jepa_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, JepaHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        for n, p in zip(JEPA_PARAM_NAMES, eng.params_list):
            small = n in JEPA_SMALL
            pv = p.detach().cpu().numpy()
            close(f"{tag} {n}", pv if small else keep(pv), g[f"{tag}_{n}"], rtol=3e-4, floor=1e-5, abs_floor=NOISE.get(n, 1e-7))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"], rtol=5e-4, atol=2e-4)


def test_full_size_batch_vs_oracle_and_indexed_store_with_cached_statistics():
    from efficient_probing_amd import functional as F_
    from efficient_probing_amd.engine import make_engine
    case = JepaCase("big", B=64, N=256, D=768, C=100, seed=3, sharp=True)
    inp = make_jepa_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = JO.make_head(case.D, case.C, case.heads)
    with torch.no_grad():
        for n, p in zip(JEPA_PARAM_NAMES, JO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want).max())))
    t = torch.from_numpy(inp["targets"]).to(DEV)
    store = torch.cat([x, tokens(case, inp["x_buf2"])], dim=0)
    stats = F_.token_stats(store, F_.JEPA_LN_EPS)
    idx = torch.randperm(store.shape[0], device=DEV)[:case.B].to(torch.int32)
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e2 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e1.train_step(store, t, lr=0.5, image_index=idx, token_stats=stats)
    e2.train_step(store[idx.long()].contiguous(), t, lr=0.5)
    assert torch.equal(e1.flat_p, e2.flat_p)
