"""dP[b, q, :] = dy[b, q-slice] . Wv[q-slice, :] for thin query slices on the bf16 matrix cores (csrc/ep_dp_slice.hip, round 6) --
the autograd of the value projection (reference poolings/ep.py:40) at the published protocol's 32 queries -- through the C ABI
(ep_project_backward) against float64, over ragged batches, every slice width the reference's shapes produce (24 / 32 / 36 columns at
D = 768 / 1024 / 1152, 12 / 16 / 8 / 4 and 40 / 48 / 64 elsewhere) and column counts that are no multiple of the kernel's chunk.  fp32 tolerance:
bf16 x3 products are fp32-accurate (<= 4e-6 of the result's scale).  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(1024, 768, 32), (1000, 1024, 32), (77, 1152, 32), (33, 384, 32), (129, 768, 64), (257, 512, 32), (64, 192, 16),
          (50, 1152, 48), (16, 64, 16), (300, 2048, 64),
          # slices of 36 .. 64 columns: two matrix K-steps (SigLIP2 SO400M at 32 queries: 1152 / 32 = 36)
          (1024, 1152, 32), (200, 384, 8), (130, 512, 8), (65, 2048, 32), (40, 320, 8)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{b}x{d}_q{q}" for b, d, q in SHAPES])
def test_dp_of_thin_slices_vs_fp64(shape):
    from efficient_probing_amd import functional as F_
    B, D, Q = shape
    Dq = D // Q
    assert Dq <= 64
    g = torch.Generator(device=DEV).manual_seed(B + D + Q)
    dy = torch.randn(B, D, device=DEV, generator=g)
    Wv = torch.randn(D, D, device=DEV, generator=g) * 0.05
    P = torch.zeros(B, Q, D, device=DEV)
    dP, _ = F_.project_backward(dy, None, P, Wv, None, need_dP=True, need_dWv=False)
    want = torch.einsum("bqc,qcd->bqd", dy.double().view(B, Q, Dq), Wv.double().view(Q, Dq, D))
    scale = float(want.abs().max())
    assert float((dP.double() - want).abs().max()) <= 4e-6 * scale
    dP2, _ = F_.project_backward(dy, None, P, Wv, None, need_dP=True, need_dWv=False)
    assert torch.equal(dP, dP2)                               # run to run: the same bits
    # with y and ML the same launch also leaves the softmax-correction rows delta[b, q] = dy_q . y_q in ML[b, q, 2]
    y = torch.randn(B, D, device=DEV, generator=g)
    ML = torch.full((B, Q, 4), 7.0, device=DEV)
    dP3, _ = F_.project_backward(dy, y, P, Wv, ML, need_dP=True, need_dWv=False)
    assert torch.equal(dP3, dP)
    want_d = (dy.double() * y.double()).view(B, Q, Dq).sum(-1)
    assert float((ML[:, :, 2].double() - want_d).abs().max()) <= 2e-6 * max(1.0, float(want_d.abs().max()))
    assert bool((ML[:, :, [0, 1, 3]] == 7.0).all())           # the other three columns are not touched
