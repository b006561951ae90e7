"""AbMILP head on the GPU: the native module (autograd path) and the fused engine (ep_abmilp_head_train_step through
the C ABI) against the golden vectors of the real reference and the CPU oracle.  Needs an MI355X (pytest -m gpu).
fp32 tolerances: forward rtol 5e-5 / atol 1e-5 (four chained D-long contractions and two softmaxes); gradients and updated parameters rtol 2e-4 with an absolute floor of
5e-5 of the tensor's scale (the contractions run over up to B*N rows in a different summation order)."""
import os

import numpy as np
import pytest
import torch

from cases import ABMILP_CASES, ABMILP_PARAM_NAMES, ABMILP_SMALL, STEP_LRS, AbmilpCase, make_abmilp_inputs, sub
from oracle import abmilp_oracle as AO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"abmilp_{case.name}.npz"))


def native_head(case, inp):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.abmilp import ABMILPHead
    pool = ABMILPHead(dim=case.D, self_attention_apply_to="both", content=case.content)
    head = torch.nn.Sequential(pool, probe_heads._batchnorm(case.D), probe_heads.Linear(case.D, case.C)).to(DEV).train()
    plist = list(head[0]._tensors()) + [head[2].weight, head[2].bias]
    with torch.no_grad():
        for n, p in zip(ABMILP_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    return head, plist


def close(name, got, want, rtol=2e-4, floor=5e-5, abs_floor=2e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


# d loss / d proj.bias is a sum of cancelling terms: a constant shift of every Xa row shifts `out` by the same vector,
# which BatchNorm removes again, so only the (small) predictor path survives -- its fp32 error is set by the size of
# the cancelling terms (~1e-2), not by the result (~1e-5).
# d loss / d b2 is exactly zero in exact arithmetic (softmax is shift invariant): both sides hold rounding noise only.
# d loss / d b1 nearly cancels the same way (a common shift of the hidden pre-activations moves every token's score
# together, which the token softmax removes to first order): entries ~1e-6 built from terms ~1e-2.
CANCELLING = {"proj_b": 1e-5, "b2": 1e-6, "b1": 4e-7}


@pytest.mark.parametrize("case", ABMILP_CASES, ids=lambda c: c.name)
def test_module_forward_backward_vs_reference(case):
    from efficient_probing_amd import functional as F_
    g, inp = load(case), make_abmilp_inputs(case)
    head, plist = native_head(case, inp)
    x, t = torch.from_numpy(inp["x_buf"]).to(DEV), torch.from_numpy(inp["targets"]).to(DEV)
    pooled, amap = head[0].forward_with_attn_map(x)
    logits = head[2](head[1](pooled))
    loss, _ = F_.cross_entropy_loss(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=5e-5, atol=1e-5)
    np.testing.assert_allclose(amap.cpu().numpy(), g["attn_map"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=2e-4, atol=5e-5)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=2e-5)
    keep = (lambda a: a) if case.full else sub
    for n, p in zip(ABMILP_PARAM_NAMES, plist):
        gr = p.grad.cpu().numpy()
        close(n, gr if n in ABMILP_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=CANCELLING.get(n, 1e-7))
        assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=2e-2 if n in CANCELLING else 2e-4,
                                                              abs=2e-5 if n in CANCELLING else 1e-9)


@pytest.mark.parametrize("case", ABMILP_CASES, ids=lambda c: c.name)
def test_engine_lars_steps_vs_reference(case):
    from efficient_probing_amd.engine import AbmilpHeadEngine, make_engine
    g, inp = load(case), make_abmilp_inputs(case)
    head, plist = native_head(case, inp)
    eng = make_engine(head, optimizer="lars", weight_decay=case.weight_decay)
    assert isinstance(eng, AbmilpHeadEngine)
    keep = (lambda a: a) if case.full else sub
    for step in range(case.steps):
        x = torch.from_numpy(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]).to(DEV)
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"]).to(DEV)
        eng.train_step(x, t, lr=STEP_LRS[step % len(STEP_LRS)])
        tag = f"lars{step + 1}"
        assert eng.read_stats()[0] == pytest.approx(float(g[f"{tag}_loss"]), rel=3e-5)
        for n, p, mu in zip(ABMILP_PARAM_NAMES, eng.params_list, eng.mu_views()):
            small = n in ABMILP_SMALL
            pv, mv = p.detach().cpu().numpy(), mu.cpu().numpy()
            # tensors whose gradient is rounding noise (CANCELLING) drift by lr * noise per step
            close(f"{tag} {n}", pv if small else keep(pv), g[f"{tag}_{n}"], rtol=3e-4 if n not in CANCELLING else 2e-3,
                  floor=1e-5 if n not in CANCELLING else 2e-4)
            close(f"{tag} mu {n}", mv if small else keep(mv), g[f"{tag}_mu_{n}"], rtol=1e-3, floor=2e-4,
                  abs_floor=CANCELLING.get(n, 1e-7))
        np.testing.assert_allclose(head[1].running_mean.cpu().numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(head[1].running_var.cpu().numpy(), g[f"{tag}_running_var"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(eng.eval_logits(torch.from_numpy(inp["x_buf"]).to(DEV)).cpu().numpy(), g["eval_logits"],
                               rtol=5e-4, atol=5e-5)
    # the reference's own evaluation mode (fp16 autocast, engine_finetune.py:131): operands and logits rounded to fp16
    # around the fused fp32 forward (engine._eval_logits_fp16_operands), pinned on the reference's fp16-autocast logits to a
    # few fp16 ulps of their scale; the fp32 evaluation above sits further away from them
    want16 = g["eval_logits_fp16_autocast"]
    ev16 = eng.eval_logits(torch.from_numpy(inp["x_buf"]).to(DEV), precision="fp16_autocast").cpu().numpy()
    ulp = 2.0 ** -11 * float(np.abs(want16).max())
    # The "sharp" fixtures (parameters scaled up until the softmax is nearly one-hot: scores of order 100) are outside what
    # operand rounding can pin: there the reference's fp16 rounding of the SCORES themselves -- an intermediate inside the
    # pooling -- changes the attention weights by tens of percent (measured: 570 fp16 ulps of the logits' scale at
    # abmilp tiny_sharp_patch, for the fp32 evaluation as well), so only the form of the result is checked for them.
    if "sharp" not in case.name:
        err16 = float(np.abs(ev16 - want16).max()) / ulp
        assert err16 <= 8, f"{case.name}: {err16:.1f} fp16 ulps of the logits' scale"
        assert (ev16.argmax(1) != want16.argmax(1)).sum() <= max(1, case.B // 32)
    assert np.array_equal(ev16, ev16.astype(np.float16).astype(np.float32))          # an fp16 result


def test_batch_that_fills_the_chip_vs_oracle_and_determinism():
    """64 images of 256 x 768: forward against the CPU oracle, engine gradients against the autograd path, and two
    identical runs bit for bit."""
    from efficient_probing_amd import functional as F_
    from efficient_probing_amd.engine import make_engine
    case = AbmilpCase("big", B=64, N=256, D=768, C=100, seed=3, sharp=True)
    inp = make_abmilp_inputs(case)
    head, plist = native_head(case, inp)
    x, t = torch.from_numpy(inp["x_buf"]).to(DEV), torch.from_numpy(inp["targets"]).to(DEV)
    with torch.no_grad():
        got = head[0](x).cpu().numpy()
    oh = AO.make_head(case.D, case.C)
    with torch.no_grad():
        for n, p in zip(ABMILP_PARAM_NAMES, AO.head_params(oh)):
            p.copy_(torch.from_numpy(inp[n]))
        want = oh[0](torch.from_numpy(inp["x_buf"])).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    loss, _ = F_.cross_entropy_loss(head(x), t)
    loss.backward()
    ref = [p.grad.clone() for p in plist]
    outs = []
    for _ in range(2):
        eng = make_engine(native_head(case, inp)[0], optimizer="sgd")
        eng.forward_backward(x, t)
        outs.append(eng.flat_g.clone())
        for n, a, p in zip(ABMILP_PARAM_NAMES, ref, eng.params_list):
            close(n, p.grad.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, floor=1e-6)
    assert torch.equal(outs[0], outs[1])


def test_weight_gradients_over_many_token_rows_vs_oracle():
    """40 images of 256 x 256 tokens: 10240 token rows put the D x D and 3D x D weight-gradient contractions on the K-split path
    (gemm_split_k: few output tiles, K >= 8192) -- every gradient against the CPU oracle's autograd."""
    from efficient_probing_amd import functional as F_
    case = AbmilpCase("rows", B=40, N=256, D=256, C=50, seed=5, sharp=True)
    inp = make_abmilp_inputs(case)
    head, plist = native_head(case, inp)
    x, t = torch.from_numpy(inp["x_buf"]).to(DEV), torch.from_numpy(inp["targets"]).to(DEV)
    loss, _ = F_.cross_entropy_loss(head(x), t)
    loss.backward()
    oh = AO.make_head(case.D, case.C)
    oparams = AO.head_params(oh)
    with torch.no_grad():
        for n, p in zip(ABMILP_PARAM_NAMES, oparams):
            p.copy_(torch.from_numpy(inp[n]))
    oh.train()
    oloss = torch.nn.functional.cross_entropy(oh(torch.from_numpy(inp["x_buf"])), torch.from_numpy(inp["targets"]))
    oloss.backward()
    assert loss.item() == pytest.approx(oloss.item(), rel=2e-5)
    for n, p, q in zip(ABMILP_PARAM_NAMES, plist, oparams):
        close(n, p.grad.cpu().numpy(), q.grad.numpy().reshape(p.shape), rtol=2e-4, floor=2e-5)
