"""The bf16 x3 weight-gradient tile (csrc/ep_wgrad3.h): C = A^T B over the batch index with both fp32 operands split into
three bf16 terms on the fly -- dWc = dlogits^T z and dWv_q = dy_q^T P_q (reference probe_heads.py:76, poolings/ep.py:40 under
autograd), stand-alone (ep_gemm_b3_kernel through ep_linear_backward / ep_project_backward) against float64, at the fp32
tolerances of the exact-f32 kernel it replaces (rtol 1e-4 of the tests that consume these gradients; the error measured
here is asserted at 4e-6 of the result's scale)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, want):
    return float((got.double() - want).abs().max() / want.abs().max())


@pytest.mark.parametrize("shape", [(1024, 768, 1000), (1025, 768, 1000), (257, 64, 10), (96, 384, 100), (64, 128, 7), (4096, 1152, 1000),
                                   (33, 768, 1000),
                                   # round 6: the 128 x 128 tile of long gradients (csrc/ep_wgrad3.h: gemm_tile_b3w; K >= 4096, outputs
                                   # >= 1024 x 1024): whole tiles, ragged edges on both sides and a ragged K
                                   (4096, 1152, 1024), (4100, 1156, 1100), (8192, 1024, 1280)],
                         ids=["c2", "ragged_rows", "tiny_classes", "k96", "k64_c7", "so400m_b4096", "k33_f32_path",
                              "wide_tile", "wide_tile_ragged", "wide_tile_k8192"])
def test_classifier_weight_gradient_vs_fp64(shape):
    from efficient_probing_amd import functional as F_
    B, Dp, C_ = shape
    g = torch.Generator(device=DEV).manual_seed(B + Dp)
    dl = torch.randn(B, C_, device=DEV, generator=g) * torch.rand(B, 1, device=DEV, generator=g)      # rows of different scale
    z = torch.randn(B, Dp, device=DEV, generator=g)
    W = torch.randn(C_, Dp, device=DEV, generator=g)
    _, dW, db = F_.linear_backward(dl, z, W, need_dz=False)
    want = dl.double().t() @ z.double()
    assert rel_err(dW, want) < 4e-6
    # (the bias gradient is a plain fp32 column sum over the B rows: its absolute error grows with the row count)
    np.testing.assert_allclose(db.cpu().numpy(), dl.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-5 * max(1.0, B / 1024))
    # accumulate: C += A^T B
    base = torch.randn_like(dW)
    acc = base.clone()
    F_.linear_backward(dl, z, W, need_dz=False, dWc=acc, dbc=torch.zeros(C_, device=DEV), accumulate=True)
    assert rel_err(acc, base.double() + want) < 4e-6
    # run to run: the same bits
    _, dW2, _ = F_.linear_backward(dl, z, W, need_dz=False)
    assert torch.equal(dW, dW2)


@pytest.mark.parametrize("shape", [(1024, 768, 8), (1025, 768, 8), (256, 4096, 8), (512, 384, 1), (130, 1024, 32)],
                         ids=["c2", "ragged", "vit7b_b256", "c1_q1", "q32"])
def test_value_projection_weight_gradient_vs_fp64(shape):
    """dWv[q Dq + m, :] = sum_b dy[b, q Dq + m] P[b, q, :]: Q batched contractions, 96 / 512 / 384 / 32 rows per query."""
    from efficient_probing_amd import functional as F_
    B, D, Q = shape
    g = torch.Generator(device=DEV).manual_seed(D + Q)
    dy = torch.randn(B, D, device=DEV, generator=g)
    y = torch.randn(B, D, device=DEV, generator=g)
    P = torch.randn(B, Q, D, device=DEV, generator=g)
    Wv = torch.randn(D, D, device=DEV, generator=g) * D ** -0.5
    ML = torch.zeros(B, Q, 4, device=DEV)
    _, dWv = F_.project_backward(dy, y, P, Wv, ML, need_dP=False)
    Dq = D // Q
    want = torch.cat([dy[:, q * Dq:(q + 1) * Dq].double().t() @ P[:, q].double() for q in range(Q)], 0)
    assert rel_err(dWv, want) < 4e-6


def test_switch_puts_the_exact_f32_kernel_back():
    """EP_GEMM_B3=0: the same gradients on v_mfma_f32_16x16x4_f32 (read once per process, hence a child)."""
    code = ("import torch; from efficient_probing_amd import functional as F_; g = torch.Generator(device='cuda').manual_seed(1);"
            "dl = torch.randn(512, 1000, device='cuda', generator=g); z = torch.randn(512, 768, device='cuda', generator=g);"
            "W = torch.randn(1000, 768, device='cuda', generator=g); _, dW, _ = F_.linear_backward(dl, z, W, need_dz=False);"
            "want = dl.double().t() @ z.double(); print(float((dW.double() - want).abs().max() / want.abs().max()))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs = {}
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, EP_GEMM_B3=v, PYTHONPATH=root), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        errs[v] = float(r.stdout.strip().splitlines()[-1])
    assert errs["0"] < 4e-6 and errs["1"] < 4e-6, errs


@pytest.mark.parametrize("shape", [(16384, 768, 3072), (16385, 1152, 1000), (8192, 3072, 768), (4096, 260, 772)],
                         ids=["fc1_rows16k", "ragged_so400m", "fc2", "ragged_k260_n772"])
def test_large_activation_times_weight_contractions_vs_fp64(shape):
    """The same tile takes the LARGE K / K (y = x W^T + b) and K / T (dx = dy W) contractions -- 512 output tiles and more:
    the dense layers of the matrix-core-bound heads over all B N token rows and the MLPs of the attention-pool heads --
    with both operands split on the fly; the 1024-row contractions of the EP step stay on the LDS-DMA ring kernels."""
    from efficient_probing_amd import functional as F_
    M, K, N_ = shape
    g = torch.Generator(device=DEV).manual_seed(M + K)
    x = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N_, K, device=DEV, generator=g) * K ** -0.5
    b = torch.randn(N_, device=DEV, generator=g)
    y = F_.linear_forward(x, W, b)
    want = x.double() @ W.double().t() + b.double()
    assert rel_err(y, want) < 4e-6
    dy = torch.randn(M, N_, device=DEV, generator=g)
    dx, dW, db = F_.linear_backward(dy, x, W)
    assert rel_err(dx, dy.double() @ W.double()) < 4e-6
    # (the weight gradient sums over M = 4096 ... 16385 rows in fp32: rounding of the running sum grows like sqrt(K),
    # for the exact-f32 instruction as well)
    assert rel_err(dW, dy.double().t() @ x.double()) < 4e-6 * max(1.0, (M / 1024) ** 0.5)
    y2 = F_.linear_forward(x, W, b)
    assert torch.equal(y, y2)
