"""Token passes across kernel families: every (D, Q) pair a registry head can produce is checked against a float64
evaluation -- forward pooled vectors / scores and the backward query gradient.  Regression for the matrix-core
kernel at odd D / 128 with more than 8 queries (its image epilogue overran the tile slot).  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(5, 256, 1152, 16), (5, 256, 1152, 12), (3, 100, 384, 12), (3, 100, 384, 16), (4, 197, 640, 16), (4, 50, 896, 9),
          (5, 197, 768, 16), (5, 197, 768, 12), (5, 256, 1024, 16), (5, 256, 1280, 16), (6, 196, 1152, 8), (3, 64, 256, 16),
          (70, 256, 1152, 16)]


@pytest.mark.parametrize("B,N,D,Q", SHAPES)
@pytest.mark.parametrize("amp", [1.0, 30.0])
def test_pool_forward_backward_vs_float64(B, N, D, Q, amp):
    from efficient_probing_amd import functional as F_
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + D + Q)
    x = torch.randn(B, N, D, device=DEV, generator=g)
    cls = torch.randn(Q, D, device=DEV, generator=g) * amp / D ** 0.5
    P, S, ML = F_.pool_forward(x, cls, 1.0)
    xd, cd = x.double(), cls.double()
    s = torch.einsum("qd,bnd->bqn", cd, xd)
    A = torch.softmax(s, -1)
    Pr = torch.einsum("bqn,bnd->bqd", A, xd)
    np.testing.assert_allclose(S.cpu().numpy(), s.cpu().numpy(), rtol=1e-5, atol=2e-5 * amp)
    np.testing.assert_allclose(P.cpu().numpy(), Pr.cpu().numpy(), rtol=1e-4, atol=5e-5)
    # backward: dcls = sum_b sum_n A (dA - delta) x with dA = dP . x, delta = dP . P
    dP = torch.randn(B, Q, D, device=DEV, generator=g)
    ML[:, :, 2] = (dP.double() * Pr).sum(-1).float()
    dcls = F_.pool_backward(x, S, ML, dP, 1.0)
    dA = torch.einsum("bqd,bnd->bqn", dP.double(), xd)
    dS = A * (dA - (dP.double() * Pr).sum(-1, keepdim=True))
    ref = torch.einsum("bqn,bnd->qd", dS, xd)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dcls.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=3e-5 * scale)
