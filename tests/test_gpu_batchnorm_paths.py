"""BatchNorm1d(affine=False) of the head (reference probe_heads.py:109-110) across its three kernel paths -- the one-launch
kernels (B <= 1024 rows in the registers of one workgroup), the two-launch chunked kernels, and the wide-tile kernels for many
rows (B > 4096: the DOLG head's token rows) -- against a float64 evaluation: output, running statistics (unbiased variance,
momentum 0.1), num_batches_tracked and the backward.  Needs an MI355X (pytest -m gpu)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-6


@pytest.mark.parametrize("B,D", [(2, 16), (7, 100), (63, 768), (64, 64), (65, 36), (1000, 768), (1024, 1152), (1025, 768), (3000, 96),
                                 (4096, 128), (4097, 768), (20000, 100)])
def test_train_forward_backward_vs_float64(B, D):
    from efficient_probing_amd import functional as F_
    g = torch.Generator(device=DEV).manual_seed(B * 31 + D)
    y = (torch.randn(B, D, device=DEV, generator=g) * (0.2 + 3.0 * torch.rand(1, D, device=DEV, generator=g))
         + 2.0 * torch.randn(1, D, device=DEV, generator=g)).requires_grad_(True)
    rm = torch.randn(D, device=DEV, generator=g) * 0.1
    rv = torch.rand(D, device=DEV, generator=g) + 0.5
    rm0, rv0 = rm.double().clone(), rv.double().clone()
    nbt = torch.zeros((), device=DEV, dtype=torch.int64)
    z = F_.batch_norm_train(y, rm, rv, nbt, EPS, 0.1)
    dz = torch.randn(B, D, device=DEV, generator=g)
    z.backward(dz)
    yd = y.detach().double()
    mu, var = yd.mean(0), yd.var(0, unbiased=False)
    rs = 1.0 / torch.sqrt(var + EPS)
    zr = (yd - mu) * rs
    np.testing.assert_allclose(z.detach().cpu().numpy(), zr.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), (0.9 * rm0 + 0.1 * mu).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 * rv0 + 0.1 * yd.var(0, unbiased=True)).cpu().numpy(), rtol=2e-5, atol=1e-6)
    assert int(nbt) == 1
    dzd = dz.double()
    dyr = rs * (dzd - dzd.mean(0) - zr * (dzd * zr).mean(0))
    scale = float(dyr.abs().max())
    np.testing.assert_allclose(y.grad.cpu().numpy(), dyr.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
