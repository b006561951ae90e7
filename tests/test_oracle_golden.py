"""Pin the numpy oracle (oracle/ep_oracle.py) against the golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from cases import EPCLS_CASES, make_epcls_inputs, CASES, LR_POINTS, STEP_LRS, make_inputs, view_tokens, sub, keeper, assert_mu_close, post_bn_tol, trust_ratio_gaps, assert_amp_bf16_fidelity
from oracle import ep_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FWD = dict(rtol=1e-5, atol=1e-6)
GRAD = dict(rtol=1e-4, atol=1e-6)


def load(case):
    return np.load(os.path.join(GOLD, f"ep_{case.name}.npz"))


def state_from(case, inp):
    Dp = case.D // case.d_out
    return O.HeadState(cls_token=inp["cls_token"].copy(), v_weight=inp["v_weight"].copy(),
                       fc_weight=inp["fc_weight"].copy(), fc_bias=inp["fc_bias"].copy(),
                       running_mean=np.zeros(Dp, np.float32), running_var=np.ones(Dp, np.float32),
                       num_queries=case.Q, d_out=case.d_out)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_forward_and_grads(case):
    g, inp = load(case), make_inputs(case)
    st = state_from(case, inp)
    x = view_tokens(case, inp["x_buf"])
    out, cache = O.head_forward_train(st, x, inp["targets"])
    np.testing.assert_allclose(out["pooled"], g["pooled"], **FWD)
    np.testing.assert_allclose(O.ep_attention(x, st.cls_token), g["attn"], **FWD)
    np.testing.assert_allclose(cache["ep"]["attn"], g["attn"], **FWD)
    tol = dict(rtol=1e-4, atol=1e-5) if case.B < 64 else post_bn_tol(case)
    np.testing.assert_allclose(out["z"], g["z"], **tol)
    np.testing.assert_allclose(out["logits"], g["logits"], **tol)
    np.testing.assert_allclose(out["loss"], g["loss"], rtol=1e-5)
    assert_amp_bf16_fidelity(out["logits"], out["loss"], g, err_msg=case.name)     # distance from the published --amp bfloat16 head
    a1, a5 = O.accuracy(out["logits"], inp["targets"])
    assert a1 == pytest.approx(float(g["acc1"])) and a5 == pytest.approx(float(g["acc5"]))
    gr = O.head_backward(st, cache)
    keep = keeper(case)
    for n in O.PARAM_ORDER:
        got = gr[n] if n in ("cls_token", "fc_bias") else keep(gr[n])
        want = g[f"grad_{n}"]
        scale = np.abs(want).max()
        np.testing.assert_allclose(got, want, rtol=GRAD["rtol"], atol=max(GRAD["atol"], 2e-5 * scale), err_msg=n)
        assert np.linalg.norm(gr[n].astype(np.float64)) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=1e-4)


@pytest.mark.parametrize("case", EPCLS_CASES, ids=lambda c: c.name)
def test_per_image_queries_forward_and_grads(case):
    """EfficientProbing.forward(x, cls=...) (reference poolings/ep.py:32-33) against the real reference: the pooled vector
    and the gradients of the per-image queries and of v.weight."""
    g = np.load(os.path.join(GOLD, f"epcls_{case.name}.npz"))
    inp = make_epcls_inputs(case)
    x = view_tokens(case, inp["x_buf"])
    out, cache = O.ep_forward(x, inp["cls_token"], inp["v_weight"], case.Q, d_out=case.d_out, cls=inp["cls"])
    np.testing.assert_allclose(out, g["pooled"], **FWD)
    dcls, dWv = O.ep_backward(inp["dy"], cache, inp["v_weight"])
    assert dcls.shape == (case.B, case.Q, case.D)
    for got, want, n in ((dcls, g["grad_cls"], "cls"), (keeper(case)(dWv), g["grad_v_weight"], "v_weight")):
        np.testing.assert_allclose(got, want, rtol=GRAD["rtol"], atol=max(GRAD["atol"], 2e-5 * np.abs(want).max()), err_msg=n)


@pytest.mark.parametrize("opt", ["lars", "sgd"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_optimizer_steps(case, opt):
    g, inp = load(case), make_inputs(case)
    if f"{opt}1_loss" not in g:
        pytest.skip("optimizer variant not recorded for this case")
    st = state_from(case, inp)
    keep = keeper(case)
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        tg = inp["targets"] if step % 2 == 0 else inp["targets2"]
        out = O.head_train_step(st, view_tokens(case, xb), tg, STEP_LRS[step % len(STEP_LRS)],
                                weight_decay=case.weight_decay, optimizer=opt)
        tag = f"{opt}{step + 1}"
        np.testing.assert_allclose(out["loss"], g[f"{tag}_loss"], rtol=2e-5)
        for n in O.PARAM_ORDER:
            got = getattr(st, n)
            got = got if n in ("cls_token", "fc_bias") else keep(got)
            np.testing.assert_allclose(got, g[f"{tag}_{n}"], rtol=1e-4, atol=2e-6, err_msg=f"{tag} {n}")
            if opt == "lars":
                mu = st.mu[n] if n in ("cls_token", "fc_bias") else keep(st.mu[n])
                want = g[f"{tag}_mu_{n}"]
                # torch's CPU float32 norm (naive per-lane accumulation over >1e6 elements) is itself off by
                # 1.5e-4 .. 9e-4 relative to the exact norm: one common factor on the trust ratio and hence on mu
                assert_mu_close(mu, want, err_msg=f"{tag} mu {n}", gaps=trust_ratio_gaps(g, n, step + 1))
        np.testing.assert_allclose(st.running_mean, g[f"{tag}_running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st.running_var, g[f"{tag}_running_var"], rtol=1e-5, atol=1e-6)
        assert st.num_batches_tracked == int(g[f"{tag}_nbt"])
    if opt == "lars":
        ev = O.head_forward_eval(st, view_tokens(case, inp["x_buf"]))
        np.testing.assert_allclose(ev, g["eval_logits"], rtol=2e-4, atol=2e-4)
        if "eval_logits_fp16_autocast" in g.files:
            # the reference's evaluation mode (fp16 autocast, engine_finetune.py:131): an fp16 result is pinned to a few
            # fp16 ulps of the logits' scale (summation order inside the fp32-accumulating matmuls differs)
            ev16 = O.head_forward_eval_fp16_autocast(st, view_tokens(case, inp["x_buf"]))
            want = g["eval_logits_fp16_autocast"]
            np.testing.assert_allclose(ev16, want, rtol=0, atol=4 * 2.0 ** -11 * float(np.abs(want).max()))
            # and it is a different thing from the fp32 evaluation: the switch matters at this resolution
            assert float(np.abs(want - g["eval_logits"]).max()) > 1e-6


def test_lr_schedule_table():
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["lr"]
    assert len(fx) == len(LR_POINTS)
    for row in fx:
        got = O.adjust_learning_rate(row["epoch"], row["lr"], row["min_lr"], row["warmup"], row["epochs"])
        assert got == pytest.approx(row["out"], rel=1e-12, abs=1e-15)
        assert row["group0"] == pytest.approx(got) and row["group1"] == pytest.approx(0.5 * got)
    assert O.absolute_lr(0.1, 4096) == pytest.approx(1.6)


def test_lars_edge_cases():
    g = np.load(os.path.join(GOLD, "lars_edges.npz"))
    ps = [g[f"p{i}_before"] for i in range(5)]
    gs = [g[f"g{i}"] for i in range(5)]
    mus = [None] * 5
    for step in (1, 2):
        ps, mus = O.lars_step(ps, gs, mus, lr=0.5)
        for i in range(5):
            np.testing.assert_allclose(ps[i], g[f"p{i}_after{step}"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(mus[i], g[f"mu{i}_after{step}"], rtol=1e-5, atol=1e-7)
    ps = [g[f"p{i}_before"] for i in range(5)]
    ps, _ = O.lars_step(ps, gs, [None] * 5, lr=0.5, weight_decay=0.01)
    for i in range(5):
        np.testing.assert_allclose(ps[i], g[f"wd_p{i}_after1"], rtol=1e-5, atol=1e-7)


def test_grad_scaler_trajectory():
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["scaler"]
    st = O.GradScalerState(growth_interval=fx["growth_interval"])
    for i, want in enumerate(fx["scale"]):
        found = i in fx["inf_at"]
        assert fx["stepped"][i] == (not found)
        st.update(found)
        assert st.scale == want


def test_refactoring_identity():
    """pool-then-project (what the HIP kernels do) == project-then-pool (reference)."""
    case = CASES[2]
    inp = make_inputs(case)
    x = view_tokens(case, inp["x_buf"])
    out, cache = O.ep_forward(x, inp["cls_token"], inp["v_weight"], case.Q)
    P = np.matmul(cache["attn"], x)                              # (B,Q,D)
    dq = case.D // case.Q
    Wq = inp["v_weight"].reshape(case.Q, dq, case.D)
    alt = np.einsum("bqd,qcd->bqc", P, Wq).reshape(case.B, -1)
    np.testing.assert_allclose(alt, out, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["tiny_q4", "tiny_q8", "vitb16_q8"])
def test_torch_port_matches_golden(name):
    """The CPU-baseline port (oracle/torch_port.py) computes what the reference computes."""
    import torch
    from cases import CASE_BY_NAME
    from oracle import torch_port as TP
    case = CASE_BY_NAME[name]
    g, inp = load(case), make_inputs(case)
    head = TP.make_head(case.D, case.Q, case.C, case.d_out).train()
    with torch.no_grad():
        head[0].cls_token.copy_(torch.from_numpy(inp["cls_token"]))
        head[0].v.weight.copy_(torch.from_numpy(inp["v_weight"]))
        head[2].weight.copy_(torch.from_numpy(inp["fc_weight"]))
        head[2].bias.copy_(torch.from_numpy(inp["fc_bias"]))
    mus = [torch.zeros_like(p) for p in head.parameters()]
    x = torch.from_numpy(view_tokens(case, inp["x_buf"]))
    loss = TP.train_step(head, mus, x, torch.from_numpy(inp["targets"]), STEP_LRS[0])
    assert float(loss) == pytest.approx(float(g["lars1_loss"]), rel=1e-6)
    keep = keeper(case)
    np.testing.assert_allclose(head[0].cls_token.detach().numpy(), g["lars1_cls_token"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(keep(head[2].weight.detach().numpy()), g["lars1_fc_weight"], rtol=1e-5, atol=1e-7)
