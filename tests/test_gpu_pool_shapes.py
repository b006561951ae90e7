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


IMGQ = [(5, 197, 768, 1), (5, 197, 768, 12), (4, 256, 1152, 12), (4, 256, 1152, 1), (3, 50, 384, 12), (3, 50, 384, 1),
        (4, 100, 1024, 4), (4, 100, 1024, 16), (2, 196, 4096, 1), (3, 33, 2048, 1), (70, 196, 768, 12), (3, 17, 256, 8),
        (3, 40, 1280, 8), (2, 20, 1536, 1)]


@pytest.mark.parametrize("B,N,D,H", IMGQ)
@pytest.mark.parametrize("mode", ["raw", "ln_scores_raw_pool", "ln"])
def test_per_image_query_passes_vs_float64(B, N, D, H, mode):
    """csrc/ep_pool_imgq.hip: per-image query rows over channel slices, per-image query gradients."""
    from efficient_probing_amd import functional as F_
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + D + H)
    x = torch.randn(B, N, D, device=DEV, generator=g) * (0.5 + torch.rand(B, N, 1, device=DEV, generator=g) * 2) \
        + torch.randn(B, N, 1, device=DEV, generator=g)
    dh = D // H
    u = torch.randn(B, D, device=DEV, generator=g) * 4.0 / dh ** 0.5
    ts = F_.token_stats(x, 1e-6) if mode != "raw" else None
    pool_ln = mode == "ln"
    P, ML = F_.imgq_pool_forward(x, u, H, ts, pool_ln)
    xd = x.double()
    xh = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-6)
    k = xd if mode == "raw" else xh
    v = xh if pool_ln else xd
    ks, vs, us = k.view(B, N, H, dh), v.view(B, N, H, dh), u.double().view(B, H, dh)
    s = torch.einsum("bhc,bnhc->bhn", us, ks)
    A = torch.softmax(s, -1)
    Pr = torch.einsum("bhn,bnhc->bhc", A, vs)
    np.testing.assert_allclose(P.cpu().numpy(), Pr.reshape(B, D).cpu().numpy(), rtol=1e-4, atol=5e-5)
    dP = torch.randn(B, D, device=DEV, generator=g)
    du = F_.imgq_pool_backward(x, u, H, P, ML, dP, ts, pool_ln)
    dPs = dP.double().view(B, H, dh)
    dA = torch.einsum("bhc,bnhc->bhn", dPs, vs)
    dS = A * (dA - (dPs * Pr).sum(-1, keepdim=True))
    ref = torch.einsum("bhn,bnhc->bhc", dS, ks).reshape(B, D)
    # (sharp softmax over scores of magnitude ~10 and |dA| ~ 100: fp32 noise of the scores alone is ~1e-4 of the result)
    np.testing.assert_allclose(du.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))


ROWQ = [(5, 256, 768, 4), (3, 197, 768, 4), (4, 49, 64, 4), (3, 100, 1152, 4), (2, 256, 1024, 3), (6, 37, 128, 1), (2, 64, 1280, 2),
        (70, 256, 768, 4)]


@pytest.mark.parametrize("B,N,D,Q", ROWQ)
@pytest.mark.parametrize("mode", ["raw", "ln", "ln_bias_bf16"])
def test_full_width_per_image_query_passes_vs_float64(B, N, D, Q, mode):
    """ep_rowq_pool_forward / backward (ep_imgqf_kernel): Q <= 4 full-width query rows PER IMAGE, optional LayerNorm-of-tokens
    mode, additive score bias, additive dA term, explicit dS, per-image query gradients -- against a float64 evaluation;
    ragged token counts (tails of the 2- / 4-token iterations and of the four token waves) and bf16 tokens included."""
    from efficient_probing_amd import functional as F_
    g = torch.Generator(device=DEV).manual_seed(B * 977 + D + 3 * Q + N)
    x = torch.randn(B, N, D, device=DEV, generator=g) * (0.5 + torch.rand(B, N, 1, device=DEV, generator=g)) + 0.3
    bf16 = mode.endswith("bf16")
    if bf16:
        x = x.to(torch.bfloat16)
    xd = x.double()
    ln = mode != "raw"
    tstat = F_.token_stats(x, 1e-6) if ln else None
    if ln:
        ts = tstat.double()
        xd = (xd - ts[..., :1]) * ts[..., 1:]
    u = torch.randn(B, Q, D, device=DEV, generator=g) * 3.0 / D ** 0.5
    sb = torch.randn(B, Q, N, device=DEV, generator=g) if "bias" in mode else None
    P, S, ML = F_.rowq_pool_forward(x, u, tstat, sb)
    s = torch.einsum("bqd,bnd->bqn", u.double(), xd) + (sb.double() if sb is not None else 0.0)
    A = torch.softmax(s, -1)
    Pr = torch.einsum("bqn,bnd->bqd", A, xd)
    np.testing.assert_allclose(S.cpu().numpy(), s.cpu().numpy(), rtol=1e-5, atol=3e-5)
    np.testing.assert_allclose(P.cpu().numpy(), Pr.cpu().numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(ML[:, :, 0].cpu().numpy(), s.max(-1).values.cpu().numpy(), rtol=1e-5, atol=3e-5)
    lsum = torch.exp(s - ML[:, :, :1].double()).sum(-1)
    np.testing.assert_allclose(ML[:, :, 1].cpu().numpy(), lsum.cpu().numpy(), rtol=2e-5)
    dP = torch.randn(B, Q, D, device=DEV, generator=g)
    db = torch.randn(B, Q, N, device=DEV, generator=g) if sb is not None else None
    dA = torch.einsum("bqd,bnd->bqn", dP.double(), xd) + (db.double() if db is not None else 0.0)
    delta = (A * dA).sum(-1)
    ML[:, :, 2] = delta.float()
    du, dS = F_.rowq_pool_backward(x, S, ML, dP, tstat, db, True)
    dSr = A * (dA - delta[..., None])
    ref = torch.einsum("bqn,bnd->bqd", dSr, xd)
    np.testing.assert_allclose(dS.cpu().numpy(), dSr.cpu().numpy(), rtol=1e-4, atol=3e-6 * float(dSr.abs().max()) + 1e-7)
    np.testing.assert_allclose(du.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=3e-5 * float(ref.abs().max()))
