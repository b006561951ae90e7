"""bf16-STORED wide rows (D = 4096: the pre-dumped DINOv3 ViT-7B tokens of BASELINE configs[4]) on the hybrid token passes of
csrc/ep_pool_wideb.hip -- scores / dA on the bf16 matrix cores against a three-term split of the fp32 operand, pooling on the
vector ALU: both passes against float64 on the stored values (reference poolings/ep.py:41-44 and its autograd), ragged last
tiles, fewer than 8 queries, a strided token view, an indexed batch of a resident store, query chunks (Q = 32).  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(9, 40, 8, 1.0), (5, 17, 3, 3.0), (3, 196, 8, 3.0), (300, 31, 8, 1.0), (4, 8, 8, 10.0), (2, 3, 1, 1.0), (260, 9, 5, 1.0)]


def _ref(x, cls, dP):
    xs = x.double()
    s = torch.matmul(cls.double(), xs.transpose(1, 2))
    A = torch.softmax(s, -1)
    P = torch.matmul(A, xs)
    dA = torch.matmul(dP.double(), xs.transpose(1, 2))
    dS = A * (dA - (dP.double() * P).sum(-1, keepdim=True))
    return s, P, torch.einsum("bqn,bnd->qd", dS, xs)


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{b}x{n}_q{q}" for b, n, q, _ in SHAPES])
def test_both_passes_match_float64(shape):
    from efficient_probing_amd import functional as F_, _native
    B, Nn, Q, amp = shape
    D = 4096
    lib = _native.load()
    assert lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 0, _native.EP_DTYPE_BF16).decode() == "ep_pool_wideb_fwd_kernel"
    assert lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 1, _native.EP_DTYPE_BF16).decode() == "ep_pool_wideb_bwd_kernel"
    g = torch.Generator(device=DEV).manual_seed(B + Nn + Q)
    buf = torch.randn(B, Nn + 1, D, device=DEV, generator=g).to(torch.bfloat16)
    x = buf[:, 1:]                                                 # the strided view models_more.py:24 produces
    cls = torch.randn(Q, D, device=DEV, generator=g) * amp / D ** 0.5
    dP = torch.randn(B, Q, D, device=DEV, generator=g)
    P, S, ML = F_.pool_forward(x, cls, 1.0)
    s, Pr, dcr = _ref(x, cls, dP)
    np.testing.assert_allclose(S.double().cpu().numpy(), s.cpu().numpy(), rtol=2e-6, atol=4e-6 * max(1.0, amp))
    assert float((P.double() - Pr).abs().max()) <= 5e-6 * float(Pr.abs().max())
    lse = ML[..., 0].double() + ML[..., 1].double().log()
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(s, -1).cpu().numpy(), rtol=1e-6, atol=4e-6 * max(1.0, amp))
    ML2 = ML.clone()
    ML2[..., 2] = (dP * P).sum(-1)
    dcls = F_.pool_backward(x, S, ML2, dP, 1.0)
    scale = float(dcr.abs().max())
    if Nn > 1:
        assert float((dcls.double() - dcr).abs().max()) <= 2e-5 * scale
    # the vector-ALU kernels of ep_pool_wide.hip on the same inputs (through the generic cross-check the library keeps)
    Pg, Sg, MLg = F_.pool_forward(x.float(), cls, 1.0)             # fp32 copy of the same values: the fp32 wide kernels
    assert float((P - Pg).abs().max()) <= 5e-6 * float(Pr.abs().max())


def test_indexed_batch_and_query_chunks():
    from efficient_probing_amd import functional as F_
    D, M, Nn = 4096, 40, 21
    g = torch.Generator(device=DEV).manual_seed(3)
    store = torch.randn(M, Nn, D, device=DEV, generator=g).to(torch.bfloat16)
    idx = torch.randint(0, M, (13,), device=DEV, generator=g, dtype=torch.int32)
    for Q in (8, 32):                                              # 32 queries: four chunks of 8 with the chunk's memory stride
        cls = torch.randn(Q, D, device=DEV, generator=g) * 2.0 / D ** 0.5
        dP = torch.randn(13, Q, D, device=DEV, generator=g)
        P, S, ML = F_.pool_forward(store, cls, 1.0, image_index=idx)
        x = store[idx.long()]
        s, Pr, dcr = _ref(x, cls, dP)
        assert float((P.double() - Pr).abs().max()) <= 5e-6 * float(Pr.abs().max())
        np.testing.assert_allclose(S.double().cpu().numpy(), s.cpu().numpy(), rtol=2e-6, atol=8e-6)
        ML2 = ML.clone()
        ML2[..., 2] = (dP * P).sum(-1)
        dcls = F_.pool_backward(store, S, ML2, dP, 1.0, image_index=idx)
        assert float((dcls.double() - dcr).abs().max()) <= 2e-5 * float(dcr.abs().max())
