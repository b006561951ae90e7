"""The AMP-bf16 arithmetic mode of the EP head's fused step (ep_head_step.arith = EP_ARITH_BF16_AUTOCAST, ABI v25;
ProbeHeadEngine(arithmetic="bf16_autocast")): the six contractions of a step as ONE bf16 matrix-core product with fp32
accumulation -- what the published runs' ``--amp bfloat16`` does inside autocast (reference engine_finetune.py:52-55) minus the
bf16 rounding of the outputs.  Pinned against the reference head's own bf16-autocast forward recorded in the EP goldens
(``logits_bf16_autocast`` / ``loss_bf16_autocast``: cases.assert_amp_bf16_fidelity), against the fp32 mode on a short
training run, and refused loudly where the contractions cannot run against the weight planes.  Needs an MI355X."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from cases import CASES, make_inputs, assert_amp_bf16_fidelity  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _capable(c):
    # round 6: every slice width -- until round 5 the mode needed (D / d_out / Q) % 32 == 0 (dP contracts over a query's slice
    # of the planes' permuted k-order); the published rows at 256 x 768, Q = 32 (slice 24) and SigLIP2 SO400M (144 / 36) were refused
    return c.D % 4 == 0 and (c.D // c.d_out) % 4 == 0


def _head(case, inp):
    from test_gpu_parity import build_head
    return build_head(case, inp)


def _tokens(case, buf):
    from test_gpu_parity import tokens
    return tokens(case, buf)


AMP_CASES = [c for c in CASES if _capable(c)]


@pytest.mark.parametrize("case", AMP_CASES, ids=lambda c: c.name)
def test_train_logits_within_the_reference_bf16_heads_distance(case):
    from efficient_probing_amd.engine import ProbeHeadEngine
    g = np.load(os.path.join(GOLD, f"ep_{case.name}.npz"))
    if "logits_bf16_autocast" not in g.files:
        pytest.skip("fixture without the bf16-autocast forward")
    inp = make_inputs(case)
    x = _tokens(case, inp["x_buf"])
    t = torch.from_numpy(inp["targets"]).to(DEV)
    out = {}
    for mode in ("fp32", "bf16_autocast"):
        eng = ProbeHeadEngine(_head(case, inp), optimizer="sgd", lr=0.0, arithmetic=mode)
        eng.train_step(x, t, lr=0.0)                               # (lr = 0: the forward of the fixture's parameters)
        loss, _, _, bad = eng.read_stats()
        assert bad == 0 and int(eng.found_inf.item()) == 0
        out[mode] = (eng.last_train_logits().cpu().numpy(), loss)
    # the fp32 mode reproduces the fixture's fp32 logits; the AMP mode is a different arithmetic ...
    np.testing.assert_allclose(out["fp32"][0], g["logits"], rtol=2e-4, atol=2e-4)
    scale = float(np.abs(g["logits"]).max())
    d = float(np.abs(out["bf16_autocast"][0] - out["fp32"][0]).max())
    assert d > 1e-5 * scale, "the AMP mode returned the fp32 logits: single-product kernels not in use?"
    # ... no further from the reference's bf16-autocast head than that head is from the reference's fp32 one
    assert_amp_bf16_fidelity(out["bf16_autocast"][0], out["bf16_autocast"][1], g, err_msg=case.name)
    assert d <= 4.0 * 2.0 ** -8 * scale


MB2_CASES = [c for c in AMP_CASES if c.D in (256, 384, 512, 768, 1024) and c.Q <= 16 and not getattr(c, "strided", False)]


@pytest.mark.parametrize("case", MB2_CASES, ids=lambda c: c.name)
def test_bf16_stored_tokens_run_single_product_passes(case):
    """bf16-STORED tokens are what autocast makes of the tokens inside the reference's matmuls, so this is the closest form of
    the published arithmetic: the token passes (csrc/ep_pool_mb.hip, mb2 kernels with NT = 1) multiply bf16(query) x token and
    bf16(weight) x token as one product as well.  Same bound against the reference's bf16-autocast head."""
    from efficient_probing_amd.engine import ProbeHeadEngine
    from efficient_probing_amd import _native
    g = np.load(os.path.join(GOLD, f"ep_{case.name}.npz"))
    if "logits_bf16_autocast" not in g.files:
        pytest.skip("fixture without the bf16-autocast forward")
    inp = make_inputs(case)
    x = _tokens(case, inp["x_buf"]).to(torch.bfloat16)
    t = torch.from_numpy(inp["targets"]).to(DEV)
    lib = _native.load()
    assert "mb2" in lib.ep_pool_kernel_name_ex(case.B, x.shape[1], case.D, case.Q, 0, _native.EP_DTYPE_BF16).decode()
    out = {}
    for mode in ("fp32", "bf16_autocast"):
        eng = ProbeHeadEngine(_head(case, inp), optimizer="sgd", lr=0.0, arithmetic=mode)
        eng.train_step(x, t, lr=0.0)
        loss, _, _, bad = eng.read_stats()
        assert bad == 0 and int(eng.found_inf.item()) == 0
        out[mode] = (eng.last_train_logits().cpu().numpy(), loss)
    scale = float(np.abs(g["logits"]).max())
    d = float(np.abs(out["bf16_autocast"][0] - out["fp32"][0]).max())
    assert 1e-5 * scale < d <= 4.0 * 2.0 ** -8 * scale, d / (2.0 ** -8 * scale)
    assert_amp_bf16_fidelity(out["bf16_autocast"][0], out["bf16_autocast"][1], g, err_msg=case.name)
    # gradients: three steps of LARS in both modes stay close (the update is a trust-ratio-normalised direction)
    e32 = ProbeHeadEngine(_head(case, inp), optimizer="lars", lr=0.1, arithmetic="fp32")
    e16 = ProbeHeadEngine(_head(case, inp), optimizer="lars", lr=0.1, arithmetic="bf16_autocast")
    for _ in range(3):
        e32.train_step(x, t, lr=0.1); e16.train_step(x, t, lr=0.1)
    for a, b in zip(e32.params_list, e16.params_list):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        assert float((a - b).norm() / a.norm()) < 2e-2


def test_wide_bf16_rows_in_the_amp_mode_against_the_vit7b_golden():
    """196 x 4096 with bf16-stored tokens: the hybrid passes (csrc/ep_pool_wideb.hip) multiply bf16(query) x token only in this
    mode; same bound against the reference's bf16-autocast head on the ViT-7B fixture."""
    from efficient_probing_amd.engine import ProbeHeadEngine
    case = [c for c in CASES if c.name == "vit7b_q8"][0]
    g = np.load(os.path.join(GOLD, f"ep_{case.name}.npz"))
    if "logits_bf16_autocast" not in g.files:
        pytest.skip("fixture without the bf16-autocast forward")
    inp = make_inputs(case)
    x = _tokens(case, inp["x_buf"]).to(torch.bfloat16)
    t = torch.from_numpy(inp["targets"]).to(DEV)
    out = {}
    for mode in ("fp32", "bf16_autocast"):
        eng = ProbeHeadEngine(_head(case, inp), optimizer="sgd", lr=0.0, arithmetic=mode)
        eng.train_step(x, t, lr=0.0)
        loss, _, _, bad = eng.read_stats()
        assert bad == 0
        out[mode] = (eng.last_train_logits().cpu().numpy(), loss)
    scale = float(np.abs(g["logits"]).max())
    d = float(np.abs(out["bf16_autocast"][0] - out["fp32"][0]).max())
    assert 1e-5 * scale < d <= 4.0 * 2.0 ** -8 * scale, d / (2.0 ** -8 * scale)
    assert_amp_bf16_fidelity(out["bf16_autocast"][0], out["bf16_autocast"][1], g, err_msg=case.name)
    e32 = ProbeHeadEngine(_head(case, inp), optimizer="lars", lr=0.1, arithmetic="fp32")
    e16 = ProbeHeadEngine(_head(case, inp), optimizer="lars", lr=0.1, arithmetic="bf16_autocast")
    for _ in range(2):
        e32.train_step(x, t, lr=0.1); e16.train_step(x, t, lr=0.1)
    for a, b in zip(e32.params_list, e16.params_list):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        assert float((a - b).norm() / a.norm()) < 2e-2


def test_short_training_run_follows_the_fp32_mode():
    """40 LARS steps at the published learning-rate scale on a learnable synthetic problem: the loss curve of the AMP mode
    stays within 2 % of the fp32 mode's at every step and ends as low."""
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine
    from argparse import Namespace
    B, Nn, D, Q, Cc = 256, 64, 256, 8, 16
    g = torch.Generator(device=DEV).manual_seed(11)
    proto = torch.randn(Cc, D, device=DEV, generator=g)
    targets = torch.randint(0, Cc, (B,), device=DEV, generator=g)
    x = torch.randn(B, Nn, D, device=DEV, generator=g) + 0.7 * proto[targets][:, None, :]
    curves = {}
    for mode in ("fp32", "bf16_autocast"):
        class Enc(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.head = torch.nn.Linear(D, Cc)
        torch.manual_seed(3)
        enc = Enc()
        probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=Cc))
        eng = ProbeHeadEngine(enc.head.to(DEV).train(), optimizer="lars", lr=0.4, arithmetic=mode)
        losses = []
        for _ in range(40):
            eng.train_step(x, targets, lr=0.4)
            loss, _, _, bad = eng.read_stats()
            assert bad == 0
            losses.append(loss)
        curves[mode] = np.array(losses)
    a, b = curves["fp32"], curves["bf16_autocast"]
    assert b[-1] < 0.5 * b[0], b
    np.testing.assert_allclose(b, a, rtol=2e-2, atol=2e-3)
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("dims", [(1152, 8), (768, 32), (1152, 32)], ids=["so400m_q8", "vitb_q32", "so400m_q32"])
def test_published_widths_whose_slices_are_no_multiple_of_32_train(dims):
    """The published rows' shapes the mode refused until round 5 (reference README.md:119-120, 639-645: --amp bfloat16 with
    --ep_queries 32 on ViT-B/14, SigLIP2 SO400M at 1152): a short LARS run follows the fp32 mode's loss curve."""
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine
    from argparse import Namespace
    D, Q = dims
    B, Nn, Cc = 64, 36, 10
    g = torch.Generator(device=DEV).manual_seed(5)
    proto = torch.randn(Cc, D, device=DEV, generator=g)
    t = torch.randint(0, Cc, (B,), device=DEV, generator=g)
    x = torch.randn(B, Nn, D, device=DEV, generator=g) + 0.7 * proto[t][:, None, :]
    curves = {}
    for mode in ("fp32", "bf16_autocast"):
        class Enc(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.head = torch.nn.Linear(D, Cc)
        torch.manual_seed(2)
        enc = Enc()
        probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=Cc))
        eng = ProbeHeadEngine(enc.head.to(DEV).train(), optimizer="lars", lr=0.4, arithmetic=mode)
        losses = []
        for _ in range(12):
            eng.train_step(x, t, lr=0.4)
            loss, _, _, bad = eng.read_stats()
            assert bad == 0 and int(eng.found_inf.item()) == 0
            losses.append(loss)
        curves[mode] = np.array(losses)
    a, b = curves["fp32"], curves["bf16_autocast"]
    assert b[-1] < 0.8 * b[0], b
    np.testing.assert_allclose(b, a, rtol=3e-2, atol=3e-3)
    assert not np.array_equal(a, b)


def test_other_heads_and_bad_names_are_refused_on_the_host():
    from efficient_probing_amd.engine import ProbeHeadEngine
    from cases import CASES as _C
    case = [c for c in _C if c.name == "tiny_q1"][0]
    inp = make_inputs(case)
    with pytest.raises(ValueError, match="arithmetic"):
        ProbeHeadEngine(_head(case, inp), optimizer="sgd", arithmetic="fp8")


# ---- round 6: the two heads BASELINE configs[3] compares EP with run the mode as well (reference engine_finetune.py:52-55 wraps
# every head in the same autocast; poolings/abmilp.py:53-71, poolings/coca_pytorch.py:250-343) ----
def _other_head(fam, case):
    if fam == "coca":
        import test_gpu_coca as T
        from cases import make_coca_inputs as mk
        from efficient_probing_amd.engine import CocaHeadEngine as E
    else:
        import test_gpu_abmilp as T
        from cases import make_abmilp_inputs as mk
        from efficient_probing_amd.engine import AbmilpHeadEngine as E
    inp = mk(case)
    x = T.tokens(case, inp["x_buf"]) if fam == "coca" else torch.from_numpy(inp["x_buf"]).to(DEV)
    return T, E, inp, x, torch.from_numpy(inp["targets"]).to(DEV)


def _other_cases():
    from cases import COCA_CASES, ABMILP_CASES
    return [("coca", c) for c in COCA_CASES] + [("abmilp", c) for c in ABMILP_CASES]


@pytest.mark.parametrize("fam_case", _other_cases(), ids=lambda fc: f"{fc[0]}_{fc[1].name}")
def test_coca_and_abmilp_heads_in_the_amp_mode(fam_case):
    fam, case = fam_case
    T, E, inp, x, t = _other_head(fam, case)
    g = T.load(case)
    if "logits_bf16_autocast" not in g.files:
        pytest.skip("fixture without the bf16-autocast forward")
    out = {}
    for mode in ("fp32", "bf16_autocast"):
        head, _ = T.native_head(case, inp)
        eng = E(head, optimizer="sgd", lr=0.0, arithmetic=mode)
        eng.train_step(x, t, lr=0.0)
        loss, _, _, bad = eng.read_stats()
        assert bad == 0 and int(eng.found_inf.item()) == 0
        out[mode] = (eng.last_train_logits().cpu().numpy(), loss)
    np.testing.assert_allclose(out["fp32"][0], g["logits"], rtol=2e-4, atol=2e-4)
    scale = float(np.abs(g["logits"]).max())
    d = float(np.abs(out["bf16_autocast"][0] - out["fp32"][0]).max())
    assert d > 1e-5 * scale, "the AMP mode returned the fp32 logits: single-product kernels not in use?"
    assert_amp_bf16_fidelity(out["bf16_autocast"][0], out["bf16_autocast"][1], g, err_msg=f"{fam}_{case.name}", slack=2.0)
    # ... and no further from the fp32 mode than the reference's bf16 head is from its fp32 one (the other leg of the triangle)
    ulp = 2.0 ** -8 * scale
    own = float(np.abs(np.asarray(g["logits"], np.float64) - np.asarray(g["logits_bf16_autocast"], np.float64)).max()) / ulp
    assert d <= max(4.0, 1.25 * own) * ulp, (d / ulp, own)
    # three LARS steps in both modes stay close
    heads = [T.native_head(case, inp)[0] for _ in range(2)]
    e32, e16 = E(heads[0], optimizer="lars", lr=0.1, arithmetic="fp32"), E(heads[1], optimizer="lars", lr=0.1, arithmetic="bf16_autocast")
    for _ in range(3):
        e32.train_step(x, t, lr=0.1); e16.train_step(x, t, lr=0.1)
    for a, b in zip(e32.params_list, e16.params_list):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        if float(a.norm()) > 0:
            # (LARS steps the 1-D tensors by lr x gradient without a trust ratio, and several of this head's bias gradients are
            # sums of cancelling terms -- tests/test_gpu_abmilp.py CANCELLING: their drift is not normalised)
            assert float((a - b).norm() / a.norm()) < (3e-2 if a.dim() > 1 else 1e-1)


# ---- round 6: the long weight gradients of the matrix-core-bound heads in the AMP mode run on the single-product form of the
# 128 x 128 tile (csrc/ep_wgrad3.h: gemm_tile_b3w1; reference poolings/abmilp.py:53-71 under autograd inside autocast) ----
_WIDE1_CHILD = r"""
import hashlib, sys
import numpy as np, torch
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/golden"]
from cases import AbmilpCase, make_abmilp_inputs
import test_gpu_abmilp as T
from efficient_probing_amd.engine import AbmilpHeadEngine as E
case = AbmilpCase("wide1", B=64, N=256, D=1024, C=10, seed=11, sharp=False)
inp = make_abmilp_inputs(case)
x, t = torch.from_numpy(inp["x_buf"]).cuda(), torch.from_numpy(inp["targets"]).cuda()
g = {}
for mode in ("fp32", "bf16_autocast"):
    eng = E(T.native_head(case, inp)[0], optimizer="sgd", lr=0.0, arithmetic=mode)
    eng.forward_backward(x, t)
    g[mode] = [p.grad.detach().double().cpu() for p in eng.params_list]
    if mode == "bf16_autocast":
        print("sha", hashlib.sha256(eng.flat_g.cpu().numpy().tobytes()).hexdigest())
for a, b in zip(g["fp32"], g["bf16_autocast"]):
    print("rel", a.dim(), float((a - b).norm() / max(float(a.norm()), 1e-30)), float(a.norm()))
"""


def test_long_weight_gradients_on_the_single_product_wide_tile():
    """64 images of 256 x 1024 tokens: 16384 token rows put dW1, dWp (1024 x 1024) and dWqkv (3072 x 1024) on the wide tile (K slices
    included), and 64 batch entries put the six per-image attention products (K/K, K/T and T/T layouts) on it as well.
    (a) the AMP-mode gradients of the matrices stay within bf16 rounding of the fp32 mode's (an indexing error is O(1); the fp32 mode
    at this batch size is pinned against the oracle in tests/test_gpu_abmilp.py);
    (b) the tile's own single-product form and the three-term tile's run-time branch (EP_B3_WIDE1=0) give the same bits:
    same rounding, same order of the matrix instructions.  (The switch is read once per process: children.)"""
    import os, subprocess, sys
    tests = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(tests)
    sha = {}
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _WIDE1_CHILD, tests], cwd=root, env=dict(os.environ, EP_B3_WIDE1=v, PYTHONPATH=root),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        sha[v] = [l for l in lines if l.startswith("sha")][0]
        rels = [l.split() for l in lines if l.startswith("rel")]
        assert len(rels) == 9
        for _, dim, rel, norm in rels:
            if int(dim) > 1 and float(norm) > 0:
                assert 1e-6 < float(rel) < 2e-2, (v, rels)
    assert sha["1"] == sha["0"], sha
