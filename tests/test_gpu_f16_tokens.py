"""fp16-STORED tokens (EP_DTYPE_F16, ABI v24): what the reference's evaluate() hands the head under its fp16 autocast
(engine_finetune.py:131).  The forward entry points of the EP head read them in place -- widened to fp32 in the token ring,
exactly -- so the results equal the fp32 path run on the widened copy up to summation order (another tile geometry);
everything that is not a forward refuses the storage type loudly.  Needs an MI355X (pytest -m gpu)."""
import ctypes as C

import numpy as np
import pytest
import torch
from argparse import Namespace

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(9, 197, 768, 8), (5, 256, 768, 32), (4, 196, 384, 1), (3, 50, 1024, 8), (6, 33, 1152, 8), (2, 20, 2048, 8), (7, 17, 200, 4)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{b}x{n}x{d}_q{q}" for b, n, d, q in SHAPES])
def test_pool_forward_reads_fp16_tokens_in_place(shape):
    from efficient_probing_amd import functional as F_, _native
    B, Nn, D, Q = shape
    lib = _native.load()
    g = torch.Generator(device=DEV).manual_seed(B + Nn + D + Q)
    buf = torch.randn(B, Nn + 1, D, device=DEV, generator=g).to(torch.float16)
    x16 = buf[:, 1:]                                               # a strided view, as models_more.py:24 produces
    cls = torch.randn(Q, D, device=DEV, generator=g) * 3.0 / D ** 0.5        # scores of order 3: a non-trivial softmax
    name = lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 0, _native.EP_DTYPE_F16).decode()
    assert name in ("ep_pool_fwd_kernel", "ep_pool_fwd_generic_kernel"), name
    if D % 8 == 0:
        assert F_.as_token_view(x16, allow_f16=True)[0].data_ptr() == x16.data_ptr()      # read in place, no copy
    if F_.f16_in_place_ok(D):
        P16, S16, ML16 = F_.pool_forward(x16, cls, 1.0)
    else:                                                          # (the Python layer widens these; the C ABI still takes them)
        xv, bs = F_.as_token_view(x16, allow_f16=True)
        P16 = torch.empty(B, Q, D, device=DEV); S16 = torch.empty(B, Q, Nn, device=DEV); ML16 = torch.empty(B, Q, 4, device=DEV)
        _native.check(lib.ep_pool_forward(xv.data_ptr(), F_.token_dtype_code(xv), bs, 0, B, Nn, D, cls.data_ptr(), 0, Q, 1.0, P16.data_ptr(),
                                          S16.data_ptr(), ML16.data_ptr(), 0, 0, _native.current_stream_ptr(torch.device(DEV))), "ep_pool_forward")
    P32, S32, ML32 = F_.pool_forward(x16.float(), cls, 1.0)       # the widened copy through the fp32 kernels
    np.testing.assert_allclose(S16.cpu().numpy(), S32.cpu().numpy(), rtol=1e-5, atol=1e-5)       # (scores of magnitude ~20: one fp32 ulp is 2e-6)
    np.testing.assert_allclose(P16.cpu().numpy(), P32.cpu().numpy(), rtol=1e-5, atol=5e-6)       # (two kernel families: summation order)
    # ML = {running max, sum of exponentials}: the running max is LAZY (moved only when a score exceeds it by 12), so only the
    # log-sum-exp max + log(sum) is the same in two kernel families
    lse = lambda ML: (ML[..., 0].double() + ML[..., 1].double().log()).cpu().numpy()
    np.testing.assert_allclose(lse(ML16), lse(ML32), rtol=1e-6, atol=2e-6)
    # float64 on the stored values
    xs = x16.double()
    s = torch.matmul(cls.double(), xs.transpose(1, 2))
    Pr = torch.matmul(torch.softmax(s, -1), xs)
    assert float((P16.double() - Pr).abs().max()) <= 5e-6 * max(1.0, float(Pr.abs().max()))


def test_eval_logits_take_fp16_tokens_and_everything_else_refuses_them():
    from efficient_probing_amd import probe_heads, functional as F_, _native
    from efficient_probing_amd.engine import ProbeHeadEngine
    B, Nn, D, Q, Cc = 24, 197, 768, 8, 100

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(D, Cc)
    torch.manual_seed(0)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=Cc))
    head = enc.head.to(DEV).train()
    eng = ProbeHeadEngine(head, optimizer="lars", lr=0.1)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, Nn, D, device=DEV, generator=g)
    t = torch.randint(0, Cc, (B,), device=DEV, generator=g)
    eng.train_step(x, t, lr=0.1)                                   # (running statistics that are not the initial ones)
    x16 = x.to(torch.float16)
    a = eng.eval_logits(x16).cpu().numpy()                         # fp16 tokens read in place by ep_head_eval_forward
    b = eng.eval_logits(x16.float()).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5)
    # the reference's evaluation mode on fp16 tokens = the same on their fp32 copy (rounding to fp16 is idempotent)
    c = eng.eval_logits(x16, precision="fp16_autocast").cpu().numpy()
    d = eng.eval_logits(x16.float(), precision="fp16_autocast").cpu().numpy()
    np.testing.assert_array_equal(c, d)
    # training on fp16-stored tokens widens them on the host (no silent mis-read as bf16 / fp32) ...
    eng.train_step(x16, t, lr=0.1)
    assert eng.read_stats()[3] == 0
    # ... and the C ABI refuses the storage type outside the forward entry points
    lib = _native.load()
    S = torch.zeros(B, Q, Nn, device=DEV); ML = torch.ones(B, Q, 4, device=DEV); dP = torch.zeros(B, Q, D, device=DEV)
    dcls = torch.zeros(Q, D, device=DEV)
    nws = lib.ep_pool_workspace_bytes(B, Nn, D, Q); ws = torch.zeros(nws, device=DEV, dtype=torch.uint8)
    rc = lib.ep_pool_backward(x16.data_ptr(), _native.EP_DTYPE_F16, Nn * D, 0, B, Nn, D, Q, 1.0, S.data_ptr(), ML.data_ptr(),
                              dP.data_ptr(), dcls.data_ptr(), 0, ws.data_ptr(), nws, _native.current_stream_ptr(torch.device(DEV)))
    assert rc < 0 and "not implemented" in _native.last_error()
