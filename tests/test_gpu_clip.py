"""CLIP attention-pooling head on the GPU: the native module (autograd path) and the fused engine (ep_clip_head_train_step
through the C ABI; token passes with per-image query rows, a position score bias and the mean row merged in as an extra softmax entry) against the golden vectors of the real reference and the
CPU oracle.  Needs an MI355X (pytest -m gpu).  fp32 tolerances: forward rtol 2e-5 / atol 1e-5 of the output scale;
gradients and updated parameters are compared with a float64 evaluation of the oracle: rtol 2e-4 with an absolute floor of
4x the real reference's own fp32 error there (golden vs float64), at least 3e-5 of the tensor's scale -- two LayerNorms and an
MLP sit behind the attention, and the rows that reach the parameters are batch-cancelling (see tests/test_cait_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from cases import CLIP_CASES, CLIP_PARAM_NAMES, CLIP_SMALL, STEP_LRS, ClipCase, make_clip_inputs, siglip_sub
from oracle import clip_oracle as CO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"clip_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.clip import AttentionPool2d
    head = torch.nn.Sequential(AttentionPool2d(in_features=case.D, feat_size=int(round(case.N ** 0.5))), probe_heads._batchnorm(case.D),
                               probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(CLIP_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    return head, plist


def tokens(case, buf):
    t = torch.from_numpy(buf).to(DEV)
    return t[:, 1:] if case.strided else t


def f64_reference(case, inp):
    """The oracle in float64 over the case's LARS steps: per step the loss, every gradient, every parameter after the update and
    the BatchNorm running statistics -- the yardstick for what fp32 can resolve here."""
    from oracle.torch_port import lars_update
    head = CO.make_head(case.D, case.C, case.N).double()
    params = CO.head_params(head)
    with torch.no_grad():
        for n, p in zip(CLIP_PARAM_NAMES, params):
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


def close_to_truth(name, got, gold, truth, rtol=2e-4, floor=3e-5):
    """`got` (the HIP path) must be as close to the float64 truth as the real reference's own fp32 result `gold` is (x4), or
    within the usual fp32 floor -- the gradients of this head are sums of batch-cancelling rows (tests/test_cait_cpu.py)."""
    scale = max(float(np.abs(truth).max()), 1e-12)
    ref_noise = float(np.abs(gold - truth).max())
    np.testing.assert_allclose(got, truth, rtol=rtol, atol=max(4.0 * ref_noise, floor * scale, 1e-7), err_msg=name)


@pytest.mark.parametrize("case", CLIP_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_clip_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    y2, attn = head[0](x, return_attn=True)
    pooled = head[0](x)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    assert torch.equal(y2, pooled.detach())
    np.testing.assert_allclose(attn.cpu().numpy(), g["attn"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=2e-5,
                               atol=1e-5 * max(1.0, float(np.abs(g["pooled"]).max())))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=1e-4)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=3e-5)
    keep = (lambda a: a) if case.full else siglip_sub
    truth = f64_reference(case, inp)[0]["grads"]
    for n, p, tr in zip(CLIP_PARAM_NAMES, plist, truth):
        gr = p.grad.cpu().numpy()
        small = n in CLIP_SMALL
        close_to_truth(n, gr if small else keep(gr), g[f"grad_{n}"], (tr if small else keep(tr)).astype(np.float64).reshape(g[f"grad_{n}"].shape))


@pytest.mark.parametrize("case", CLIP_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import ClipHeadEngine, make_engine
    g, inp = load(case), make_clip_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, ClipHeadEngine)
    keep = (lambda a: a) if case.full else siglip_sub
    truth = f64_reference(case, inp)
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=5e-5)
        for n, p, tr in zip(CLIP_PARAM_NAMES, eng.params_list, truth[step]["params"]):
            small = n in CLIP_SMALL
            pv = p.detach().cpu().numpy()
            gold = g[f"{tag}_{n}"]
            close_to_truth(f"{tag} {n}", pv if small else keep(pv), gold, (tr if small else keep(tr)).reshape(gold.shape), rtol=3e-4,
                           floor=1e-5)
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=2e-4, atol=5e-6)
    close_to_truth("eval logits", eng.eval_logits(tokens(case, inp["x_buf"])).cpu().numpy(), g["eval_logits"],
                   truth[-1]["eval_logits"], rtol=5e-4, floor=1e-4)


def test_full_size_batch_vs_oracle_and_indexed_store_with_cached_statistics():
    from efficient_probing_amd import functional as F_
    from efficient_probing_amd.engine import make_engine
    case = ClipCase("big", B=64, N=256, D=768, C=100, seed=3, sharp=True)
    inp = make_clip_inputs(case)
    head, plist = native_head(case, inp)
    x = tokens(case, inp["x_buf"])
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = CO.make_head(case.D, case.C, case.N)
    with torch.no_grad():
        for n, p in zip(CLIP_PARAM_NAMES, CO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=4e-5 * max(1.0, float(np.abs(want).max())))
    # a resident store: statistics computed ONCE for the store, batches drawn by index -- equal to gathered batches
    t = torch.from_numpy(inp["targets"]).to(DEV)
    store = torch.cat([x, tokens(case, inp["x_buf2"])], dim=0)
    stats = F_.token_stats(store, F_.CLIP_LN_EPS)
    idx = torch.randperm(store.shape[0], device=DEV)[:case.B].to(torch.int32)
    e1 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e2 = make_engine(native_head(case, inp)[0], optimizer="lars")
    e1.train_step(store, t, lr=0.5, image_index=idx, token_stats=stats)
    e2.train_step(store[idx.long()].contiguous(), t, lr=0.5)
    assert torch.equal(e1.flat_p, e2.flat_p)


@pytest.mark.parametrize("shape", [(7, 36, 64, False), (5, 196, 768, False), (3, 100, 1152, True), (4, 256, 768, True), (6, 64, 128, False), (2, 144, 1024, False)],
                         ids=lambda s: "B%d_N%d_D%d%s" % (s[0], s[1], s[2], "_bf16" if s[3] else ""))
def test_full_width_per_image_query_kernel_vs_generic_kernels(shape):
    """The token passes of this head run on ep_imgqf_kernel (every token read once for all four heads); the generic kernels
    (one read per head) are the independent implementation of the same contract: pooled rows, scores, softmax statistics,
    per-image query gradients and explicit dS must agree to fp32 summation-order noise, ragged N and bf16 tokens included."""
    from efficient_probing_amd import _native as N_
    from efficient_probing_amd import functional as F_
    B, Nn, D, bf16 = shape
    case = ClipCase("k", B=B, N=Nn, D=D, C=10, seed=11, sharp=True)
    inp = make_clip_inputs(case)
    lib = N_.load()
    outs = []
    for mode in (0, 1):
        lib.ep_debug_force_generic_pool(mode)
        try:
            head, plist = native_head(case, inp)
            x = tokens(case, inp["x_buf"])
            x = x.to(torch.bfloat16) if bf16 else x
            y = head[0](x)
            (y * torch.linspace(-1.0, 1.0, y.numel(), device=DEV).view_as(y)).sum().backward()
            outs.append([y.detach().float().cpu().numpy()] + [p.grad.cpu().numpy() for p in plist[:-2]])
        finally:
            lib.ep_debug_force_generic_pool(0)
    for i, (a, g) in enumerate(zip(*outs)):
        # with bf16 tokens the module returns the pooled rows in bf16: values 1e-7 apart can round to neighbouring bf16 numbers
        rtol = 1e-2 if (bf16 and i == 0) else 2e-4
        np.testing.assert_allclose(a, g, rtol=rtol, atol=2e-5 * max(1e-3, float(np.abs(g).max())))


@pytest.mark.parametrize("B", [96, 640], ids=["rows384", "rows2560_k_slices"])
def test_gradients_of_a_batch_with_many_image_head_rows_vs_float64(B):
    """B = 96 images of 8 x 8 tokens: 384 (image, head) rows put the column reductions of the backward on several row chunks --
    every gradient against the oracle in float64 (floor 1e-4 of the tensor's scale: the gradients of this head are sums of
    batch-cancelling rows).  B = 640 (round 6): 2560 rows put the two position-embedding gradients -- two 64 x 64 tiles summed over
    all rows -- on K slices (csrc/ep_gemm.hip: gemm_split_k, the few-tile rule)."""
    from efficient_probing_amd import functional as F_
    case = ClipCase("rows", B=B, N=64, D=128, C=20, seed=9, sharp=True)
    inp = make_clip_inputs(case)
    head, plist = native_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"]).to(DEV)
    loss, _ = F_.cross_entropy_loss(head(x), t)
    loss.backward()
    truth = f64_reference(ClipCase("rows", B=B, N=64, D=128, C=20, seed=9, sharp=True, steps=1), inp)[0]
    assert loss.item() == pytest.approx(truth["loss"], rel=2e-5)
    for n, p, tr in zip(CLIP_PARAM_NAMES, plist, truth["grads"]):
        scale = max(float(np.abs(tr).max()), 1e-12)
        np.testing.assert_allclose(p.grad.cpu().numpy(), tr.reshape(p.shape), rtol=3e-4, atol=max(1e-4 * scale, 1e-7), err_msg=n)
