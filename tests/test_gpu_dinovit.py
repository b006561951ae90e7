"""DINOv2-block head on the GPU: the native module (autograd path) and the fused engine (ep_dinovit_head_train_step through the
C ABI; every contraction of the block on the exact-fp32 matrix-core kernel) against the golden vectors of the real reference and
the CPU oracle.  Needs an MI355X (pytest -m gpu).  fp32 tolerances: attention rtol 2e-4, forward rtol 2e-5 / atol 1e-5 of the
output scale; gradients and updated parameters are compared with a float64 evaluation of the oracle: rtol 2e-4 with an absolute
floor of 4x the real reference's own fp32 error there (golden vs float64), at least 3e-5 of the tensor's scale -- a whole
transformer block (two LayerNorms, softmax attention and a GELU MLP) sits between the parameters and the loss."""
import os

import numpy as np
import pytest
import torch

from cases import (DINOVIT_ATTN_ROWS, DINOVIT_CASES, DINOVIT_PARAM_NAMES, DINOVIT_SMALL, STEP_LRS, DinovitCase, make_dinovit_inputs,
                   siglip_sub)
from oracle import dinovit_oracle as DO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"dinovit_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.dinovit import DinoViTBlockPooling
    head = torch.nn.Sequential(DinoViTBlockPooling(d_model=case.D), probe_heads._batchnorm(case.D),
                               probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(DINOVIT_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def f64_reference(case, inp):
    """The oracle in float64 over the case's LARS steps: per step the loss, every gradient and every parameter after the
    update -- the yardstick for what fp32 can resolve here."""
    from oracle.torch_port import lars_update
    head = DO.make_head(case.D, case.C).double()
    params = DO.head_params(head)
    with torch.no_grad():
        for n, p in zip(DINOVIT_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]).double())
    head.train()
    mus = [torch.zeros_like(p) for p in params]
    rec = []
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        x = torch.from_numpy(np.ascontiguousarray(xb[:, 1:] if case.strided else xb)).double()
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(head(x), t)
        loss.backward()
        grads = [p.grad.numpy().copy() for p in params]
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        rec.append(dict(loss=loss.item(), grads=grads, params=[p.detach().numpy().copy() for p in params]))
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        rec[-1]["eval_logits"] = head(torch.from_numpy(np.ascontiguousarray(xb[:, 1:] if case.strided else xb)).double()).numpy()
    return rec


# d loss / d mlp.fc2.bias = sum_b dout[b] is exactly zero in exact arithmetic (dout comes out of the BatchNorm1d backward, whose
# columns sum to zero): rounding noise of the size of one ulp of the summands on both sides
NOISE = {"fc2_b": 1e-6}


def close_to_truth(name, got, gold, truth, rtol=2e-4, floor=3e-5, abs_floor=1e-7):
    """`got` (the HIP path) must be as close to the float64 truth as the real reference's own fp32 result `gold` is (x4), or
    within the usual fp32 floor."""
    scale = max(float(np.abs(truth).max()), 1e-12)
    ref_noise = float(np.abs(gold - truth).max())
    np.testing.assert_allclose(got, truth, rtol=rtol, atol=max(4.0 * ref_noise, floor * scale, abs_floor), err_msg=name)


@pytest.mark.parametrize("case", DINOVIT_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_dinovit_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    y2, attn = head[0](x, return_attention=True)
    pooled = head[0](x)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    assert torch.equal(y2, pooled.detach())
    a = attn.cpu().numpy()
    np.testing.assert_allclose(a if case.full else a[:, :, ::DINOVIT_ATTN_ROWS], g["attn"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=2e-5,
                               atol=1e-5 * max(1.0, float(np.abs(g["pooled"]).max())))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=1e-4)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=3e-5)
    keep = (lambda a: a) if case.full else siglip_sub
    truth = f64_reference(case, inp)[0]["grads"]
    for n, p, tr in zip(DINOVIT_PARAM_NAMES, plist, truth):
        gr = p.grad.cpu().numpy()
        small = n in DINOVIT_SMALL
        close_to_truth(n, gr if small else keep(gr), g[f"grad_{n}"], (tr if small else keep(tr)).astype(np.float64).reshape(g[f"grad_{n}"].shape),
                       abs_floor=NOISE.get(n, 1e-7))


@pytest.mark.parametrize("case", DINOVIT_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import DinovitHeadEngine, make_engine
    g, inp = load(case), make_dinovit_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, DinovitHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    truth = f64_reference(case, inp)
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        for n, p, tr in zip(DINOVIT_PARAM_NAMES, eng.params_list, truth[step]["params"]):
            small = n in DINOVIT_SMALL
            pv = p.detach().cpu().numpy()
            gold = g[f"{tag}_{n}"]
            close_to_truth(f"{tag} {n}", pv if small else keep(pv), gold, (tr if small else keep(tr)).reshape(gold.shape), rtol=3e-4,
                           floor=1e-5)
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=2e-4, atol=5e-6)
    close_to_truth("eval logits", eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"],
                   truth[-1]["eval_logits"], rtol=5e-4, floor=1e-4)


def test_larger_batch_vs_oracle_and_determinism():
    """B = 32 images of 256 x 768 tokens (8192 token rows through every contraction, 256 batched attention products): pooled
    output against the fp32 oracle; the fused step is deterministic; an indexed batch equals the gathered one."""
    case = DinovitCase("big", B=32, N=256, D=768, C=100, seed=3, sharp=True)
    inp = make_dinovit_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = DO.make_head(case.D, case.C)
    with torch.no_grad():
        for n, p in zip(DINOVIT_PARAM_NAMES, DO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=4e-5 * max(1.0, float(np.abs(want).max())))
    from efficient_probing_amd.engine import make_engine
    t = torch.from_numpy(inp["targets"]).to(DEV)
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e2 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e1.train_step(x, t, lr=0.5)
    e2.train_step(x.clone(), t, lr=0.5)
    assert torch.equal(e1.flat_p, e2.flat_p)                       # deterministic: same inputs -> same bits
    # a batch of a resident store (image_index): this matrix-core-bound head gathers it into a contiguous tensor -- same bits
    perm = torch.randperm(case.B, device=DEV).to(torch.int32)
    e1.train_step(x, t[perm.long()], lr=0.5, image_index=perm)
    e2.train_step(x[perm.long()].contiguous(), t[perm.long()], lr=0.5)
    assert torch.equal(e1.flat_p, e2.flat_p)
