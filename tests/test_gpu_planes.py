"""Pre-split weight planes (csrc/ep_planes.hip): the three bf16 terms of every weight reproduce the fp32 value exactly,
and the contraction against the planes equals the fp32 contraction (checked against float64) at the error of an fp32
fmaf chain.  These are the kernels the fused train step runs its four critical-path contractions on
(reference poolings/ep.py:40, probe_heads.py:76 and their autograd).  Needs an MI355X (pytest -m gpu)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def unpermute(planes, rows, K):
    """int16 planes [3][rows][Kp] (k permuted inside groups of 32) -> float64 (3, rows, K) in natural k order."""
    Kp = (K + 31) // 32 * 32
    t = planes.view(3, rows, Kp).to(torch.int32)
    vals = ((t & 0xffff) << 16).view(torch.float32) if False else (t << 16).view(torch.float32)
    pos = torch.arange(Kp)
    p = pos % 32
    kk, g, j = p // 8, (p % 8) // 4, p % 4
    k_of_pos = (pos // 32) * 32 + 16 * g + 4 * kk + j            # position 8 kk + 4 g + j holds k = 16 g + 4 kk + j
    out = torch.zeros(3, rows, Kp, dtype=torch.float64)
    out[:, :, k_of_pos] = vals.double().cpu()
    return out[:, :, :K], out[:, :, K:]


@pytest.mark.parametrize("shape", [(768, 768), (1000, 768), (70, 100), (33, 36), (64, 64), (1, 4)], ids=lambda s: "x".join(map(str, s)))
def test_three_terms_are_exact_in_both_orientations(shape):
    from efficient_probing_amd import functional as F_
    R, K = shape
    g = torch.Generator().manual_seed(R * 1000 + K)
    W = (torch.randn(R, K, generator=g) * torch.exp(3 * torch.randn(R, K, generator=g))).to(DEV)     # many magnitudes
    pn, pt = F_.planes_split(W, natural=True, transposed=True)
    terms, pad = unpermute(pn.cpu(), R, K)
    assert torch.equal(terms.sum(0), W.double().cpu())            # h + m + l == w exactly
    assert float(pad.abs().max()) == 0.0 if pad.numel() else True # zero padding up to a multiple of 32
    # each term is a bf16 value of decreasing size
    assert float((terms[1].abs() - terms[0].abs() * 2.0 ** -8).clamp_min(0).max()) == 0.0
    assert float((terms[2].abs() - terms[0].abs() * 2.0 ** -16).clamp_min(0).max()) == 0.0
    tterms, tpad = unpermute(pt.cpu(), K, R)
    assert torch.equal(tterms.sum(0), W.double().cpu().t())
    assert float(tpad.abs().max()) == 0.0 if tpad.numel() else True


@pytest.mark.parametrize("shape", [(1024, 768, 1000), (1024, 1000, 768), (130, 772, 1004), (37, 100, 52), (5, 36, 8),
                                   (64, 32, 64), (200, 96, 768), (1025, 4096, 132),
                                   (1024, 520, 4000)],      # enough 64 x 128 tiles for the wide form, ragged K and N
                         ids=lambda s: "x".join(map(str, s)))
def test_matmul_against_planes_matches_float64(shape):
    """M x K activations times (N x K weights)^T, with a bias: error relative to the largest output entry at the level
    of the fp32 kernel (tools/gemm_fuzz.py: 1.2e-6 at K = 768), for ragged M / N / K (K tail, row and column edges)."""
    from efficient_probing_amd import functional as F_
    M, K, Nw = shape
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(Nw, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nw, generator=g).to(DEV)
    pn, _ = F_.planes_split(W)
    got = F_.matmul_planes(A, pn, Nw, bias=b)
    ref = A.double() @ W.double().t() + b.double()
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    f32 = F_.linear_forward(A, W, b)
    err32 = float((f32.double() - ref).abs().max() / ref.abs().max())
    assert err < 3e-6 and err < 3 * err32 + 2e-7, (err, err32)
    # the transposed planes give the other contraction: A' (M x Nw) times W (Nw x K) = A' (W^T)^T
    _, pt = F_.planes_split(W, natural=False, transposed=True)
    A2 = torch.randn(M, Nw, generator=g).to(DEV)
    got2 = F_.matmul_planes(A2, pt, K)
    ref2 = A2.double() @ W.double()
    assert float((got2.double() - ref2).abs().max() / ref2.abs().max()) < 3e-6


@pytest.mark.parametrize("mode", ["1", "2", "auto4096"], ids=["all_four", "classifier_only", "default_at_d4096"])
def test_fused_step_with_and_without_planes_agree(mode):
    """The same three LARS steps with the planes path (EP_GEMM_PLANES=1: all four contractions; =2: logits and dz; the
    default at D = 4096 with B >= 128: all six, weight gradients included, on 64 x 128 tiles where they fill the chip) and
    without it are compared through the engine's public results: losses to 1e-6 relative, parameters to fp32
    summation-order noise."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from argparse import Namespace
from efficient_probing_amd import probe_heads
from efficient_probing_amd.engine import ProbeHeadEngine
D, B, N = (4096, 256, 24) if os.environ.get("EP_TEST_BIG") else (768, 96, 50)
class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.head = torch.nn.Linear(D, 1000)
torch.manual_seed(0); enc = Enc()
probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=8, d_out=1, nb_classes=1000))
eng = ProbeHeadEngine(enc.head.to("cuda:0").train(), optimizer="lars", lr=0.4, weight_decay=1e-4)
g = torch.Generator().manual_seed(5)
losses = []
for s in range(3):
    x = torch.randn(B, N, D, generator=g).to("cuda:0"); t = torch.randint(0, 1000, (B,), generator=g).to("cuda:0")
    eng.train_step(x, t); losses.append(eng.read_stats()[0])
torch.save({"loss": losses, "p": eng.flat_p.cpu()}, sys.argv[1])
'''
    import tempfile
    outs = []
    for flag in (mode, "0"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            env = dict(os.environ, EP_GEMM_PLANES=flag)
            if mode == "auto4096":
                env["EP_TEST_BIG"] = "1"
                if flag != "0":
                    del env["EP_GEMM_PLANES"]                     # the library's own choice at D >= 2048
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            outs.append(torch.load(f.name))
    a, b = outs
    assert np.allclose(a["loss"], b["loss"], rtol=2e-6)
    assert not torch.equal(a["p"], b["p"])                       # two different kernels really ran
    assert torch.allclose(a["p"], b["p"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("head", ["abmilp", "dinovit", "dolg"])
def test_matrix_core_heads_keep_their_f32_contractions(head):
    """The AbMILP / DINOv2-block / DOLG heads run their contractions through a weight matrix on the planes kernel by default
    (that is what their own test files exercise); EP_ABMILP_PLANES=0 / EP_DINOVIT_PLANES=0 / EP_DOLG_PLANES=0 puts them back
    on the f32 kernel, and the same golden comparisons must hold there."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{f"EP_{head.upper()}_PLANES": "0"})
    r = subprocess.run([sys.executable, "-m", "pytest", f"tests/test_gpu_{head}.py", "-x", "-q", "-m", "gpu",
                        "-k", "module_forward_backward or engine_lars_steps"], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_big_tile_kernel_in_fresh_process():
    """csrc/ep_planes_big.hip (128 x 128 tiles, taken on its own only by long contractions with >= 192 tiles) forced for EVERY
    planes contraction (EP_PLANES_BIG=1, read once per process): the ragged shapes of the matmul test (row / column edges, K tail,
    fewer K-tiles than ring stages) and the fused-step comparisons run through it in a fresh interpreter."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, EP_PLANES_BIG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "matmul_against_planes or fused_step_with_and_without", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
