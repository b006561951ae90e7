"""bf16 token storage on the GPU: the token passes read bf16 tokens in place, widen them to fp32 on the fly and compute
in fp32, so the result must equal the fp32 path run on the same (bf16-rounded) values -- to rounding of the summation
order only, and bit for bit where the kernel family and tile order are the same.  Needs an MI355X (pytest -m gpu)."""
import numpy as np
import pytest
import torch

from cases import CASE_BY_NAME, Case, make_inputs
from oracle import ep_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(6, 17, 64, 4), (5, 50, 256, 8), (9, 197, 768, 8), (300, 64, 384, 1), (7, 196, 1024, 8), (4, 256, 1152, 8),
          (3, 33, 2048, 8), (3, 20, 4096, 8), (4, 30, 200, 5)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_bf16_tokens_equal_fp32_math_on_rounded_values(shape):
    from efficient_probing_amd import functional as F_
    B, Nn, D, Q = shape
    gen = torch.Generator(device="cpu").manual_seed(4)
    xb = torch.randn(B, Nn, D, generator=gen).to(torch.bfloat16).to(DEV)
    xf = xb.float()                                            # the same values, stored as fp32
    cls = (torch.randn(Q, D, generator=gen) * 0.4).to(DEV)
    dP = torch.randn(B, Q, D, generator=gen).to(DEV)
    scale = D ** -0.5
    Pb, Sb, MLb = F_.pool_forward(xb, cls, scale)
    Pf, Sf, MLf = F_.pool_forward(xf, cls, scale)
    np.testing.assert_allclose(Sb.cpu().numpy(), Sf.cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(Pb.cpu().numpy(), Pf.cpu().numpy(), rtol=2e-5, atol=2e-6)
    ML2 = MLf.clone(); ML2[:, :, 2] = 0.3
    gb = F_.pool_backward(xb, Sf, ML2, dP, scale)
    gf = F_.pool_backward(xf, Sf, ML2, dP, scale)
    np.testing.assert_allclose(gb.cpu().numpy(), gf.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(gf.abs().max()))
    # a strided view ([:, 1:]) and an indexed resident store are read in place as well
    if D % 8 == 0:
        big = torch.randn(B + 2, Nn + 1, D, generator=gen).to(torch.bfloat16).to(DEV)
        idx = torch.randperm(B + 2, generator=gen)[:B].to(torch.int32).to(DEV)
        P1, _, _ = F_.pool_forward(big[:, 1:], cls, scale)                            # batch-strided bf16 view
        P2, _, _ = F_.pool_forward(big[:, 1:].float().contiguous(), cls, scale)
        np.testing.assert_allclose(P1.cpu().numpy(), P2.cpu().numpy(), rtol=2e-5, atol=2e-6)
        P3, _, _ = F_.pool_forward(big, cls, scale, image_index=idx)                  # in-place indexed batch
        P4, _, _ = F_.pool_forward(big[idx.long()].contiguous(), cls, scale)
        assert torch.equal(P3, P4)


def test_engine_on_bf16_tokens_matches_oracle_on_rounded_values():
    """Whole train steps on bf16 tokens against the numpy oracle fed with the same bf16-rounded values."""
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine
    from argparse import Namespace
    case = Case("bf16", B=16, N=50, D=256, Q=8, C=33, seed=12, weight_decay=1e-3)
    inp = make_inputs(case)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(case.D, case.C)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=case.Q, d_out=1, nb_classes=case.C))
    head = enc.head
    with torch.no_grad():
        head[0].cls_token.copy_(torch.from_numpy(inp["cls_token"])); head[0].v.weight.copy_(torch.from_numpy(inp["v_weight"]))
        head[2].weight.copy_(torch.from_numpy(inp["fc_weight"])); head[2].bias.copy_(torch.from_numpy(inp["fc_bias"]))
    head = head.to(DEV).train()
    eng = ProbeHeadEngine(head, optimizer="lars", weight_decay=case.weight_decay)
    st = O.HeadState(cls_token=inp["cls_token"].copy(), v_weight=inp["v_weight"].copy(), fc_weight=inp["fc_weight"].copy(),
                     fc_bias=inp["fc_bias"].copy(), running_mean=np.zeros(case.D, np.float32),
                     running_var=np.ones(case.D, np.float32), num_queries=case.Q, d_out=1)
    for step in range(3):
        xb = torch.from_numpy(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]).to(torch.bfloat16)
        tg = inp["targets"] if step % 2 == 0 else inp["targets2"]
        ref = O.head_train_step(st, xb.float().numpy(), tg, lr=0.3, weight_decay=case.weight_decay)
        eng.train_step(xb.to(DEV), torch.from_numpy(tg).to(DEV), lr=0.3)
        assert eng.read_stats()[0] == pytest.approx(float(ref["loss"]), rel=5e-5)
    for n, p in zip(O.PARAM_ORDER, eng.params_list):
        np.testing.assert_allclose(p.detach().cpu().numpy(), getattr(st, n), rtol=2e-4, atol=1e-5, err_msg=n)


def test_bf16_resident_store_trains_in_place(tmp_path):
    from efficient_probing_amd import probe_heads, token_store as TS
    from efficient_probing_amd.engine import make_engine
    from argparse import Namespace
    rng = np.random.default_rng(5)
    Nn, D, C, n = 20, 128, 10, 96
    w = TS.TokenStoreWriter(str(tmp_path), num_tokens=Nn, dim=D, shard_images=40, dtype="bfloat16")
    toks = rng.standard_normal((n, Nn, D), dtype=np.float32); labs = rng.integers(0, C, n)
    w.add(toks, labs); w.close()
    store = TS.ResidentTokenStore(str(tmp_path), DEV)
    assert store.tokens.dtype == torch.bfloat16 and store.num_images == n

    def make():
        class Enc(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.head = torch.nn.Linear(D, C)
        torch.manual_seed(0)
        enc = Enc()
        probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=4, d_out=1, nb_classes=C))
        return make_engine(enc.head.to(DEV).train(), optimizer="lars")
    e1, e2 = make(), make()
    for tokens, idx, targets in store.batches(32, epoch=0):
        e1.train_step(tokens, targets, lr=0.5, image_index=idx)
        e2.train_step(tokens[idx.long()].float().contiguous(), targets, lr=0.5)     # gathered fp32 copy
    np.testing.assert_allclose(e1.flat_p.cpu().numpy(), e2.flat_p.cpu().numpy(), rtol=1e-5, atol=1e-6)


MB_SHAPES = [(5, 50, 256, 8), (9, 197, 768, 8), (600, 256, 768, 8), (7, 196, 1024, 8), (3, 31, 512, 16), (4, 64, 768, 1),
             (2, 257, 768, 12), (5, 196, 384, 4), (4, 256, 1152, 8), (3, 40, 1024, 12), (3, 100, 256, 16), (3, 65, 512, 5),
             (3, 5, 256, 2), (2, 16, 768, 8), (1, 1, 512, 1), (1300, 33, 384, 8), (2, 3, 1152, 3)]


@pytest.mark.parametrize("shape", MB_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_bf16_matrix_core_pass_against_float64(shape):
    """ep_pool_mb_*_kernel / ep_pool_mb2_*_kernel (bf16 tokens on the bf16 matrix cores, the fp32 operand split into three bf16 terms) against a
    float64 evaluation of reference poolings/ep.py:35-44 and of its gradient on the same stored values: the split keeps
    the fp32 contract (same tolerances as the fp32 token passes), partial last tiles, N % 4 != 0, more images than
    workgroups, the 8- and 12-wave forms, the packed (Q <= 8) and the plain score exchange and up to 16 query rows
    included."""
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    lib = _native.load()
    fam = "mb2" if D in (256, 384, 512, 768, 1024) else "mb"       # 4-wave workgroups on 16-token tiles / 8-12 waves on 32
    assert lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 0, 1).decode() == f"ep_pool_{fam}_fwd_kernel"
    assert lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 1, 1).decode() == f"ep_pool_{fam}_bwd_kernel"
    gen = torch.Generator(device="cpu").manual_seed(11)
    xb = torch.randn(B, Nn, D, generator=gen).to(torch.bfloat16)
    cls = torch.randn(Q, D, generator=gen) * 0.5
    dP = torch.randn(B, Q, D, generator=gen)
    scale = D ** -0.5
    x64 = xb.double()
    S64 = torch.einsum("qd,bnd->bqn", cls.double() * scale, x64)
    A64 = torch.softmax(S64, dim=-1)
    P64 = torch.einsum("bqn,bnd->bqd", A64, x64)
    P, S, ML = F_.pool_forward(xb.to(DEV), cls.to(DEV), scale)
    np.testing.assert_allclose(S.cpu().numpy(), S64.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(P.cpu().numpy(), P64.numpy(), rtol=1e-5, atol=1e-6)
    lse = torch.logsumexp(S64, dim=-1)
    got_lse = (ML[:, :, 0].double() + torch.log(ML[:, :, 1].double())).cpu()
    np.testing.assert_allclose(got_lse.numpy(), lse.numpy(), rtol=1e-6, atol=1e-6)
    # backward: dcls = scale * sum_b,n dS[b,q,n] x[b,n,:], dS = A (dA - delta), dA = dP . x_n, delta = sum_n A dA
    dA64 = torch.einsum("bqd,bnd->bqn", dP.double(), x64)
    delta64 = (A64 * dA64).sum(-1)
    dS64 = A64 * (dA64 - delta64[..., None])
    g64 = scale * torch.einsum("bqn,bnd->qd", dS64, x64)
    ML2 = ML.clone(); ML2[:, :, 2] = delta64.float().to(DEV)
    g = F_.pool_backward(xb.to(DEV), S, ML2, dP.to(DEV), scale)
    # (noise floor: with one token the gradient is exactly zero and what is left is the fp32 rounding of delta)
    floor = 1e-6 * scale * B * float(dA64.abs().max()) * float(x64.abs().max())
    np.testing.assert_allclose(g.cpu().numpy(), g64.numpy(), rtol=1e-4, atol=2e-5 * float(g64.abs().max()) + floor)
    # the vector-ALU / generic kernels on the same stored values agree to summation order
    old = lib.ep_debug_force_generic_pool(1)
    try:
        Pg, Sg, _ = F_.pool_forward(xb.to(DEV), cls.to(DEV), scale)
    finally:
        lib.ep_debug_force_generic_pool(old)
    np.testing.assert_allclose(S.cpu().numpy(), Sg.cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(P.cpu().numpy(), Pg.cpu().numpy(), rtol=2e-5, atol=2e-6)
