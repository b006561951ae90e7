"""LayerNorm-of-tokens mode of the token passes (ep_token_stats / ep_pool_forward_ln / ep_pool_backward_ln): pooling
the normalised tokens without materialising them must equal the plain passes run on torch.layer_norm(x) -- for every
kernel family that implements the mode (vector-ALU streaming, generic).  Needs an MI355X (pytest -m gpu)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(6, 17, 64, 4), (5, 50, 256, 8), (9, 197, 768, 8), (300, 33, 384, 1), (7, 196, 1024, 8), (4, 256, 1152, 8),
          (4, 30, 200, 5), (3, 20, 4096, 8), (5, 64, 768, 16),
          # more than 8 queries: LayerNorm-of-tokens mode of the all-matrix-core kernel (ragged last tile, several images per workgroup)
          (5, 50, 768, 16), (3, 45, 1152, 16), (3, 37, 512, 12), (300, 21, 256, 9)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_ln_mode_equals_plain_passes_on_normalised_tokens(shape):
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    gen = torch.Generator(device="cpu").manual_seed(6)
    # per-token scale in [0.5, 3.5] and a per-token offset: the mode evaluates rstd * (q.x - mean * sum(q)), whose
    # rounding error grows with |mean| / std of a token, so the offsets stay comparable to the spread (as in ViT tokens)
    x = (torch.randn(B, Nn, D, generator=gen) * (0.5 + 3 * torch.rand(B, Nn, 1, generator=gen))
         + 0.5 * torch.randn(B, Nn, 1, generator=gen)).to(DEV)
    cls = (torch.randn(Q, D, generator=gen) * 0.3).to(DEV)
    dP = torch.randn(B, Q, D, generator=gen).to(DEV)
    scale, eps = D ** -0.5, 1e-5
    xhat = torch.nn.functional.layer_norm(x, (D,), eps=eps)
    stats = F_.token_stats(x, eps)
    mu, var = x.double().mean(-1), x.double().var(-1, unbiased=False)
    np.testing.assert_allclose(stats[..., 0].cpu().numpy(), mu.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(stats[..., 1].cpu().numpy(), (var + eps).rsqrt().cpu().numpy(), rtol=2e-5)
    Pr, Sr, MLr = F_.pool_forward(xhat.contiguous(), cls, scale)                     # reference: plain pass on LN(x)
    MLr2 = MLr.clone(); MLr2[:, :, 2] = (dP * Pr).sum(-1)
    gr = F_.pool_backward(xhat.contiguous(), Sr, MLr2, dP, scale)
    lib = _native.load()
    for mode in (0, 1):                                                                # automatic, generic only
        lib.ep_debug_force_generic_pool(mode)
        try:
            P, S, ML = F_.pool_forward_ln(x, cls, scale, stats)
            ML2 = ML.clone(); ML2[:, :, 2] = (dP * P).sum(-1)
            g = F_.pool_backward_ln(x, S, ML2, dP, scale, stats)
        finally:
            lib.ep_debug_force_generic_pool(0)
        np.testing.assert_allclose(S.cpu().numpy(), Sr.cpu().numpy(), rtol=2e-5, atol=3e-5, err_msg=f"S mode {mode}")
        np.testing.assert_allclose(P.cpu().numpy(), Pr.cpu().numpy(), rtol=2e-5, atol=3e-5, err_msg=f"P mode {mode}")
        np.testing.assert_allclose(g.cpu().numpy(), gr.cpu().numpy(), rtol=1e-4, atol=5e-5 * float(gr.abs().max()),
                                   err_msg=f"dcls mode {mode}")


def test_ln_mode_with_an_indexed_store():
    from efficient_probing_amd import functional as F_
    gen = torch.Generator(device="cpu").manual_seed(7)
    store = torch.randn(40, 33, 256, generator=gen).to(DEV)
    idx = torch.randperm(40, generator=gen)[:24].to(torch.int32).to(DEV)
    cls = (torch.randn(8, 256, generator=gen) * 0.3).to(DEV)
    stats = F_.token_stats(store)                                                     # once per store
    P1, S1, _ = F_.pool_forward_ln(store, cls, 0.0625, stats, image_index=idx)
    sub = store[idx.long()].contiguous()
    P2, S2, _ = F_.pool_forward_ln(sub, cls, 0.0625, F_.token_stats(sub))
    assert torch.equal(P1, P2) and torch.equal(S1, S2)


@pytest.mark.parametrize("shape", [(5, 50, 256, 8), (9, 197, 768, 8), (300, 33, 384, 1), (7, 196, 1024, 8), (5, 64, 768, 16),
                                   (6, 256, 768, 4)], ids=lambda s: "x".join(map(str, s)))
def test_ln_mode_on_bf16_stored_tokens(shape):
    """The same mode on bf16-STORED tokens (statistics and passes read the bf16 values in place, fp32 arithmetic): equal to
    the plain fp32 passes on layer_norm of the same stored values, for the streaming kernels and the generic ones."""
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    gen = torch.Generator(device="cpu").manual_seed(16)
    xb = (torch.randn(B, Nn, D, generator=gen) * (0.5 + 3 * torch.rand(B, Nn, 1, generator=gen))
          + 0.5 * torch.randn(B, Nn, 1, generator=gen)).to(torch.bfloat16).to(DEV)
    cls = (torch.randn(Q, D, generator=gen) * 0.3).to(DEV)
    dP = torch.randn(B, Q, D, generator=gen).to(DEV)
    scale, eps = D ** -0.5, 1e-5
    xhat = torch.nn.functional.layer_norm(xb.float(), (D,), eps=eps)
    stats = F_.token_stats(xb, eps)
    Pr, Sr, MLr = F_.pool_forward(xhat.contiguous(), cls, scale)
    MLr2 = MLr.clone(); MLr2[:, :, 2] = (dP * Pr).sum(-1)
    gr = F_.pool_backward(xhat.contiguous(), Sr, MLr2, dP, scale)
    lib = _native.load()
    for mode in (0, 1):
        lib.ep_debug_force_generic_pool(mode)
        try:
            P, S, ML = F_.pool_forward_ln(xb, cls, scale, stats)
            ML2 = ML.clone(); ML2[:, :, 2] = (dP * P).sum(-1)
            g = F_.pool_backward_ln(xb, S, ML2, dP, scale, stats)
        finally:
            lib.ep_debug_force_generic_pool(0)
        np.testing.assert_allclose(S.cpu().numpy(), Sr.cpu().numpy(), rtol=2e-5, atol=3e-5, err_msg=f"S mode {mode}")
        np.testing.assert_allclose(P.cpu().numpy(), Pr.cpu().numpy(), rtol=2e-5, atol=3e-5, err_msg=f"P mode {mode}")
        np.testing.assert_allclose(g.cpu().numpy(), gr.cpu().numpy(), rtol=1e-4, atol=5e-5 * float(gr.abs().max()),
                                   err_msg=f"dcls mode {mode}")
