"""CBAM pooling head on the GPU: the native module (autograd path) and the fused engine (ep_cbam_head_train_step
through the C ABI; streaming passes over the tokens + a 7x7 convolution and a one-channel BatchNorm2d on the token-grid maps) against the golden vectors of the real reference and the
CPU oracle.  Needs an MI355X (pytest -m gpu).  fp32 tolerances: forward rtol 2e-5 / atol 1e-5 of the output scale;
gradients and updated parameters rtol 2e-4 with an absolute floor of 5e-5 of the tensor's scale."""
import os

import numpy as np
import pytest
import torch

from cases import CBAM_CASES, CBAM_PARAM_NAMES, CBAM_SMALL, STEP_LRS, CbamCase, make_cbam_inputs, siglip_sub
from oracle import cbam_oracle as AO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NOISE = {}


def load(case):
    return np.load(os.path.join(GOLD, f"cbam_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.cbam import CbamPooling
    head = torch.nn.Sequential(CbamPooling(channels=case.D), probe_heads._batchnorm(case.D),
                               probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(CBAM_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
        head[0].spatial.conv.bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        head[0].spatial.conv.bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def close(name, got, want, rtol=2e-4, floor=5e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", CBAM_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_cbam_inputs(case)
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
    # the token BatchNorm's buffers moved exactly once (attention() has no side effects)
    np.testing.assert_allclose(head[0].spatial.conv.bn.running_mean.cpu().numpy(), g["lars1_tok_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(head[0].spatial.conv.bn.running_var.cpu().numpy(), g["lars1_tok_running_var"], rtol=2e-5, atol=1e-6)
    assert int(head[0].spatial.conv.bn.num_batches_tracked) == 1
    keep = (lambda a: a) if case.full else siglip_sub
    for n, p in zip(CBAM_PARAM_NAMES, plist):
        gr = p.grad.cpu().numpy()
        close(n, gr if n in CBAM_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        if n not in NOISE:
            assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=3e-4, abs=1e-9)


@pytest.mark.parametrize("case", CBAM_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import CbamHeadEngine, make_engine
    g, inp = load(case), make_cbam_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, CbamHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        for n, p, mu in zip(CBAM_PARAM_NAMES, eng.params_list, eng.mu_views()):
            small = n in CBAM_SMALL
            pv, mv = p.detach().cpu().numpy(), mu.cpu().numpy()
            close(f"{tag} {n}", pv if small else keep(pv), g[f"{tag}_{n}"], rtol=3e-4, floor=1e-5, abs_floor=NOISE.get(n, 1e-7))
            close(f"{tag} mu {n}", mv if small else keep(mv), g[f"{tag}_mu_{n}"], rtol=1e-3, floor=2e-4, abs_floor=NOISE.get(n, 1e-7))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=2e-4, atol=5e-6)
        np.testing.assert_allclose(head[0].spatial.conv.bn.running_mean.cpu().numpy(), g[f"{tag}_tok_running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(head[0].spatial.conv.bn.running_var.cpu().numpy(), g[f"{tag}_tok_running_var"], rtol=2e-5, atol=1e-6)
        assert int(head[0].spatial.conv.bn.num_batches_tracked) == int(g[f"{tag}_tok_nbt"])
    np.testing.assert_allclose(eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"], rtol=5e-4, atol=1e-4)


def test_full_size_batch_vs_oracle_indexed_store_bf16_and_determinism():
    from efficient_probing_amd import functional as F_
    from efficient_probing_amd.engine import make_engine
    case = CbamCase("big", B=64, N=256, D=768, C=100, seed=3, sharp=True)
    inp = make_cbam_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = AO.make_head(case.D, case.C)
    with torch.no_grad():
        for n, p in zip(CBAM_PARAM_NAMES, AO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        oh[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        oh[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
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
    # a resident store: the channel table computed ONCE for the store, batches drawn by index -- equal to gathered batches
    store = torch.cat([x, tokens(case, inp["x_buf2"])], dim=0)
    tab = F_.cbam_channel_table(store)
    idx = torch.randperm(store.shape[0], device=DEV)[:case.B].to(torch.int32)
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars"); e1.train_step(store, t, lr=0.5, image_index=idx, image_stats=tab)
    e2 = make_engine(native_head(case, inp)[0], optimizer="lars"); e2.train_step(store[idx.long()].contiguous(), t, lr=0.5)
    e3 = make_engine(native_head(case, inp)[0], optimizer="lars"); e3.train_step(store, t, lr=0.5, image_index=idx)
    assert torch.equal(e1.flat_p, e2.flat_p) and torch.equal(e3.flat_p, e2.flat_p)
    # bf16 token storage: equal to the fp32 path run on the rounded values
    xb = x.to(torch.bfloat16)
    e16 = make_engine(native_head(case, inp)[0], optimizer="lars"); e16.train_step(xb, t, lr=0.3)
    e32 = make_engine(native_head(case, inp)[0], optimizer="lars"); e32.train_step(xb.float(), t, lr=0.3)
    np.testing.assert_allclose(e16.flat_p.cpu().numpy(), e32.flat_p.cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_gradients_of_a_batch_that_spans_several_workgroups_vs_oracle():
    """B = 96 images of 8 x 8 tokens: the one-channel BatchNorm2d (6144 values -> three workgroups), its backward and the 7 x 7
    weight gradient (six image chunks) leave their single-workgroup forms -- every gradient against the oracle's autograd in
    float64."""
    from efficient_probing_amd import functional as F_
    case = CbamCase("wide", B=96, N=64, D=128, C=20, seed=8, sharp=True)
    inp = make_cbam_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    loss, _ = F_.cross_entropy_loss(head(x), t)
    loss.backward()
    oh = AO.make_head(case.D, case.C).double()
    oparams = AO.head_params(oh)
    with torch.no_grad():
        for n, p in zip(CBAM_PARAM_NAMES, oparams):
            p.copy_(torch.from_numpy(inp[n]).double())
        oh[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]).double())
        oh[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]).double())
    oh.train()
    oloss = torch.nn.functional.cross_entropy(oh(torch.from_numpy(inp["x_buf"]).double()), torch.from_numpy(inp["targets"]))
    oloss.backward()
    assert loss.item() == pytest.approx(oloss.item(), rel=2e-5)
    for n, p, q in zip(CBAM_PARAM_NAMES, plist, oparams):
        close(n, p.grad.cpu().numpy(), q.grad.numpy().reshape(p.shape), rtol=3e-4, floor=1e-4)
    nbn = head[0].spatial.conv.bn
    np.testing.assert_allclose(nbn.running_mean.cpu().numpy(), oh[0].bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nbn.running_var.cpu().numpy(), oh[0].bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
