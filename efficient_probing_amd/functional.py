"""Torch-facing wrappers over the C ABI (``include/ep_hip.h``): tensors in, tensors out.

Each function enqueues HIP kernels on torch's current stream and returns without
synchronising.  Tensors must be CUDA (ROCm) tensors; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _native as N

BN_EPS = 1e-6          # probe_heads.py:109-110 of the reference
BN_MOMENTUM = 0.1


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    N.require_gpu_tensor(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def as_token_view(x: torch.Tensor, allow_f16: bool = False) -> Tuple[torch.Tensor, int]:
    """Return (x', batch_stride) with x' fp32 or bf16, inner two dims contiguous.  A view such as
    ``feat[:, 1:]`` (reference models_more.py:24) is passed through without a copy.  bf16 tokens (a bf16
    backbone's output, or a bf16 token store) are read as they are -- the token passes widen them to fp32 on the
    fly and compute in fp32.  fp16 tokens (what the reference's ``evaluate()`` hands the head under its fp16 autocast,
    engine_finetune.py:131) are read as they are by the FORWARD entry points of the EP head (``allow_f16``: ABI v24,
    ``EP_DTYPE_F16``); everywhere else they are widened once here."""
    N.require_gpu_tensor(x, "tokens")
    if x.dim() != 3:
        raise ValueError(f"tokens must be (B, N, D), got {tuple(x.shape)}")
    B, Nn, D = x.shape
    if x.dtype in (torch.bfloat16, torch.float16) and D % 8 != 0:
        x = x.float()
    if x.dtype == torch.float16 and not allow_f16:
        x = x.float()
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        x = x.float()
    al = 8 if x.dtype in (torch.bfloat16, torch.float16) else 4
    ok = x.stride(2) == 1 and x.stride(1) == D and (B == 1 or x.stride(0) >= Nn * D) \
        and x.stride(0) % al == 0 and x.data_ptr() % 16 == 0
    if not ok:
        x = x.contiguous()
    return x, (x.stride(0) if B > 1 else Nn * D)


def f16_in_place_ok(D: int) -> bool:
    """fp16-stored tokens are worth reading in place where the streaming forward kernel takes the row width (D = 64 k <= 1536);
    wider rows would fall to the library's generic kernel -- slower than one widening copy in front of the wide-row kernels."""
    return D % 64 == 0 and D <= 1536


def token_dtype_code(x: torch.Tensor) -> int:
    return N.EP_DTYPE_BF16 if x.dtype == torch.bfloat16 else N.EP_DTYPE_F16 if x.dtype == torch.float16 else N.EP_DTYPE_F32


def _index_arg(image_index, x):
    """(pointer, batch size) for the optional in-place batch selection from a resident token store."""
    if image_index is None:
        return 0, x.shape[0]
    if image_index.dtype != torch.int32 or not image_index.is_cuda or not image_index.is_contiguous():
        raise ValueError("image_index must be a contiguous int32 CUDA tensor")
    return image_index.data_ptr(), image_index.numel()


def pool_forward(x: torch.Tensor, cls_token: torch.Tensor, scale: float,
                 per_image_queries: bool = False, image_index: Optional[torch.Tensor] = None):
    """EP pooling forward.  x (B,N,D); cls_token (Q,D) / (1,Q,D) or, with
    ``per_image_queries``, (B,Q,D).  Returns P (B,Q,D), S (B,Q,N), ML (B,Q,4).
    With ``image_index`` (int32, (B,)), x is a resident token store (M,N,D) and image b of the batch
    is x[image_index[b]], read in place."""
    lib = N.load()
    x, bstride = as_token_view(x, allow_f16=not per_image_queries and f16_in_place_ok(x.shape[-1]))   # (fp16 tokens: shared queries, forward only)
    _, Nn, D = x.shape
    iptr, B = _index_arg(image_index, x)
    cls = _f32c(cls_token, "cls_token")
    Q = cls.shape[-2]
    cls_bstride = Q * D if per_image_queries else 0
    P = torch.empty((B, Q, D), device=x.device, dtype=torch.float32)
    S = torch.empty((B, Q, Nn), device=x.device, dtype=torch.float32)
    ML = torch.empty((B, Q, 4), device=x.device, dtype=torch.float32)
    rc = lib.ep_pool_forward(x.data_ptr(), token_dtype_code(x), bstride, iptr, B, Nn, D, cls.data_ptr(), cls_bstride, Q,
                             float(scale), P.data_ptr(), S.data_ptr(), ML.data_ptr(), 0, 0,
                             N.current_stream_ptr(x.device))
    N.check(rc, "ep_pool_forward")
    return P, S, ML


def pool_backward(x: torch.Tensor, S: torch.Tensor, ML: torch.Tensor, dP: torch.Tensor, scale: float,
                  dcls: Optional[torch.Tensor] = None, accumulate: bool = False,
                  image_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = N.load()
    x, bstride = as_token_view(x)
    _, Nn, D = x.shape
    iptr, B = _index_arg(image_index, x)
    Q = S.shape[1]
    dP = _f32c(dP, "dP")
    if dcls is None:
        dcls = torch.empty((Q, D), device=x.device, dtype=torch.float32)
        accumulate = False
    nbytes = lib.ep_pool_workspace_bytes(B, Nn, D, Q)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    rc = lib.ep_pool_backward(x.data_ptr(), token_dtype_code(x), bstride, iptr, B, Nn, D, Q, float(scale), S.data_ptr(),
                              ML.data_ptr(), dP.data_ptr(), dcls.data_ptr(), int(accumulate), ws.data_ptr(), nbytes,
                              N.current_stream_ptr(x.device))
    N.check(rc, "ep_pool_backward")
    return dcls


def pool_backward_per_image(x: torch.Tensor, S: torch.Tensor, ML: torch.Tensor, dP: torch.Tensor, scale: float,
                            image_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradient of PER-IMAGE query rows (the ``cls=`` override of reference poolings/ep.py:32-33): (B, Q, D), one
    gradient per image -- the second token pass without the sum over the batch."""
    lib = N.load()
    x, bstride = as_token_view(x)
    _, Nn, D = x.shape
    iptr, B = _index_arg(image_index, x)
    Q = S.shape[1]
    dP = _f32c(dP, "dP")
    dq = torch.empty((B, Q, D), device=x.device, dtype=torch.float32)
    N.check(lib.ep_pool_backward_per_image(x.data_ptr(), token_dtype_code(x), bstride, iptr, B, Nn, D, Q, float(scale),
                                           S.data_ptr(), ML.data_ptr(), dP.data_ptr(), dq.data_ptr(),
                                           N.current_stream_ptr(x.device)), "ep_pool_backward_per_image")
    return dq


def attention_from_scores(S: torch.Tensor, ML: torch.Tensor) -> torch.Tensor:
    lib = N.load()
    B, Q, Nn = S.shape
    A = torch.empty_like(S)
    N.check(lib.ep_attention_from_scores(S.data_ptr(), ML.data_ptr(), B, Q, Nn, A.data_ptr(),
                                         N.current_stream_ptr(S.device)), "ep_attention_from_scores")
    return A


def project_forward(P: torch.Tensor, Wv: torch.Tensor) -> torch.Tensor:
    lib = N.load()
    B, Q, D = P.shape
    Wv = _f32c(Wv, "v.weight")
    Dp = Wv.shape[0]
    y = torch.empty((B, Dp), device=P.device, dtype=torch.float32)
    N.check(lib.ep_project_forward(P.data_ptr(), Wv.data_ptr(), B, D, Dp, Q, y.data_ptr(),
                                   N.current_stream_ptr(P.device)), "ep_project_forward")
    return y


def project_backward(dy: torch.Tensor, y: Optional[torch.Tensor], P: torch.Tensor, Wv: torch.Tensor,
                     ML: Optional[torch.Tensor], need_dP: bool = True, dWv: Optional[torch.Tensor] = None,
                     accumulate: bool = False, need_dWv: bool = True):
    lib = N.load()
    B, Q, D = P.shape
    Wv = _f32c(Wv, "v.weight")
    dy = _f32c(dy, "dy")
    Dp = Wv.shape[0]
    dP = torch.empty_like(P) if need_dP else None
    if need_dWv and dWv is None:
        dWv = torch.empty_like(Wv)
        accumulate = False
    N.check(lib.ep_project_backward(dy.data_ptr(), _ptr(y), P.data_ptr(), Wv.data_ptr(), B, D, Dp, Q, _ptr(dP),
                                    _ptr(dWv) if need_dWv else 0, _ptr(ML), int(accumulate),
                                    N.current_stream_ptr(P.device)), "ep_project_backward")
    return dP, dWv


def bn_forward_train(y, running_mean, running_var, num_batches_tracked, eps=BN_EPS, momentum=BN_MOMENTUM):
    lib = N.load()
    y = _f32c(y, "y")
    B, Dp = y.shape
    z = torch.empty_like(y)
    rstd = torch.empty((Dp,), device=y.device, dtype=torch.float32)
    ws = torch.empty(lib.ep_bn_workspace_bytes(B, Dp), device=y.device, dtype=torch.uint8)
    N.check(lib.ep_bn_forward_train(y.data_ptr(), B, Dp, eps, momentum, z.data_ptr(), rstd.data_ptr(),
                                    running_mean.data_ptr(), running_var.data_ptr(), _ptr(num_batches_tracked),
                                    ws.data_ptr(), ws.numel(), N.current_stream_ptr(y.device)), "ep_bn_forward_train")
    return z, rstd


def bn_forward_eval(y, running_mean, running_var, eps=BN_EPS):
    lib = N.load()
    y = _f32c(y, "y")
    B, Dp = y.shape
    z = torch.empty_like(y)
    N.check(lib.ep_bn_forward_eval(y.data_ptr(), B, Dp, eps, running_mean.data_ptr(), running_var.data_ptr(),
                                   z.data_ptr(), N.current_stream_ptr(y.device)), "ep_bn_forward_eval")
    return z


def bn_backward(dz, z, rstd):
    lib = N.load()
    dz = _f32c(dz, "dz")
    B, Dp = z.shape
    dy = torch.empty_like(z)
    ws = torch.empty(lib.ep_bn_workspace_bytes(B, Dp), device=z.device, dtype=torch.uint8)
    N.check(lib.ep_bn_backward(dz.data_ptr(), z.data_ptr(), rstd.data_ptr(), B, Dp, dy.data_ptr(),
                               ws.data_ptr(), ws.numel(), N.current_stream_ptr(z.device)), "ep_bn_backward")
    return dy


def padded_ld(C_: int) -> int:
    return (C_ + 3) // 4 * 4


def linear_forward(z, Wc, bc) -> torch.Tensor:
    """Returns logits as a (B, C) view of a (B, ldl) buffer (ldl = C rounded up to 4)."""
    lib = N.load()
    z = _f32c(z, "z")
    Wc = _f32c(Wc, "weight")
    B, Dp = z.shape
    C_ = Wc.shape[0]
    ldl = padded_ld(C_)
    buf = torch.empty((B, ldl), device=z.device, dtype=torch.float32)
    N.check(lib.ep_linear_forward(z.data_ptr(), Wc.data_ptr(), _ptr(bc), B, Dp, C_, buf.data_ptr(), ldl,
                                  N.current_stream_ptr(z.device)), "ep_linear_forward")
    return buf[:, :C_]


def planes_split(W: torch.Tensor, natural: bool = True, transposed: bool = False):
    """bf16 planes (three exact terms per element, csrc/ep_planes.hip) of a row-major fp32 matrix W (R, K) and / or of
    its transpose: int16 tensors for ``matmul_planes``."""
    lib = N.load()
    W = _f32c(W, "W")
    R, K = W.shape
    pn = torch.empty(lib.ep_planes_elems(R, K), device=W.device, dtype=torch.int16) if natural else None
    pt = torch.empty(lib.ep_planes_elems(K, R), device=W.device, dtype=torch.int16) if transposed else None
    N.check(lib.ep_planes_split(W.data_ptr(), R, K, K, _ptr(pn), _ptr(pt), N.current_stream_ptr(W.device)), "ep_planes_split")
    return pn, pt


def matmul_planes(A: torch.Tensor, planes: torch.Tensor, rows_w: int, bias: Optional[torch.Tensor] = None,
                  n_out: Optional[int] = None) -> torch.Tensor:
    """A (M, K) fp32 times the transpose of the (rows_w, K) matrix whose planes are given -> (M, n_out or rows_w)."""
    lib = N.load()
    A = _f32c(A, "A")
    M, K = A.shape
    n = rows_w if n_out is None else n_out
    ldc = padded_ld(n)
    buf = torch.empty((M, ldc), device=A.device, dtype=torch.float32)
    N.check(lib.ep_matmul_planes(A.data_ptr(), K, planes.data_ptr(), rows_w, K, _ptr(bias), M, n, buf.data_ptr(), ldc,
                                 N.current_stream_ptr(A.device)), "ep_matmul_planes")
    return buf[:, :n]


def _padded_rows(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """(B, C) tensor -> storage with a leading dimension that is a multiple of 4 and zero pad."""
    B, C_ = t.shape
    ldl = padded_ld(C_)
    if t.stride(1) == 1 and t.stride(0) == ldl and t.dtype == torch.float32 and ldl == C_:
        return t, ldl
    buf = torch.zeros((B, ldl), device=t.device, dtype=torch.float32)
    buf[:, :C_] = t
    return buf, ldl


def linear_backward(dlogits, z, Wc, need_dz=True, dWc=None, dbc=None, accumulate=False):
    lib = N.load()
    dl, ldl = _padded_rows(dlogits)
    z = _f32c(z, "z")
    Wc = _f32c(Wc, "weight")
    B, Dp = z.shape
    C_ = Wc.shape[0]
    dz = torch.empty_like(z) if need_dz else None
    if dWc is None:
        dWc = torch.empty_like(Wc); dbc = torch.empty((C_,), device=z.device, dtype=torch.float32)
        accumulate = False
    N.check(lib.ep_linear_backward(dl.data_ptr(), ldl, z.data_ptr(), Wc.data_ptr(), B, Dp, C_, _ptr(dz),
                                   dWc.data_ptr(), _ptr(dbc), int(accumulate),
                                   N.current_stream_ptr(z.device)), "ep_linear_backward")
    return dz, dWc, dbc


def cross_entropy(logits, targets, grad_scale: float = 1.0, need_grad: bool = True, stats=None):
    """Mean CE + accuracy counts.  Returns (row_stats (B,4), dlogits (B,C) view or None, stats (4,));
    row_stats[b] = [loss_b / B, top-1 hit, top-5 hit, non-finite]."""
    lib = N.load()
    lg, ldl = _padded_rows(logits) if not (logits.stride(1) == 1 and logits.stride(0) % 4 == 0
                                           and logits.dtype == torch.float32) else (logits, logits.stride(0))
    B, C_ = logits.shape
    targets = targets.to(device=logits.device, dtype=torch.int64).contiguous()
    loss_rows = torch.empty((B, 4), device=logits.device, dtype=torch.float32)
    dl = torch.empty((B, ldl), device=logits.device, dtype=torch.float32) if need_grad else None
    if stats is None:
        stats = torch.zeros((4,), device=logits.device, dtype=torch.float32)
    N.check(lib.ep_cross_entropy(lg.data_ptr(), ldl, targets.data_ptr(), B, C_, float(grad_scale),
                                 loss_rows.data_ptr(), _ptr(dl), stats.data_ptr(),
                                 N.current_stream_ptr(logits.device)), "ep_cross_entropy")
    return loss_rows, (dl[:, :C_] if dl is not None else None), stats


# --------------------------------------------------------------------------------------------
# optimizers on flat buffers
# --------------------------------------------------------------------------------------------
def make_segments(entries: Sequence[Tuple[int, int, bool]]):
    arr = (N.EPSegment * len(entries))()
    for i, (off, numel, trust) in enumerate(entries):
        arr[i].offset = off; arr[i].numel = numel; arr[i].apply_trust = 1 if trust else 0
    return arr


def optim_workspace(total: int, nseg: int, device) -> torch.Tensor:
    nbytes = N.load().ep_optim_workspace_bytes(total, nseg)
    return torch.empty(nbytes, device=device, dtype=torch.uint8)


def lars_step(params, grads, mu, segments, lr, weight_decay=0.0, momentum=0.9, trust_coefficient=0.001,
              inv_scale=1.0, found_inf=None, grad_norm=None, workspace=None):
    lib = N.load()
    total = params.numel()
    if found_inf is None:
        found_inf = torch.zeros((1,), device=params.device, dtype=torch.int32)
    if workspace is None:
        workspace = optim_workspace(total, len(segments), params.device)
    N.check(lib.ep_lars_step(params.data_ptr(), grads.data_ptr(), mu.data_ptr(), total, segments, len(segments),
                             lr, weight_decay, momentum, trust_coefficient, inv_scale, found_inf.data_ptr(),
                             _ptr(grad_norm), workspace.data_ptr(), workspace.numel(),
                             N.current_stream_ptr(params.device)), "ep_lars_step")
    return found_inf


def sgd_step(params, grads, lr, weight_decay=0.0, inv_scale=1.0, found_inf=None, grad_norm=None, workspace=None):
    lib = N.load()
    total = params.numel()
    if found_inf is None:
        found_inf = torch.zeros((1,), device=params.device, dtype=torch.int32)
    if workspace is None:
        workspace = optim_workspace(total, 1, params.device)
    N.check(lib.ep_sgd_step(params.data_ptr(), grads.data_ptr(), total, lr, weight_decay, inv_scale,
                            found_inf.data_ptr(), _ptr(grad_norm), workspace.data_ptr(), workspace.numel(),
                            N.current_stream_ptr(params.device)), "ep_sgd_step")
    return found_inf


def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
               inv_scale=1.0, found_inf=None, grad_norm=None, workspace=None):
    lib = N.load()
    total = params.numel()
    if found_inf is None:
        found_inf = torch.zeros((1,), device=params.device, dtype=torch.int32)
    if workspace is None:
        workspace = optim_workspace(total, 1, params.device)
    N.check(lib.ep_adamw_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), total,
                              int(step), lr, betas[0], betas[1], eps, weight_decay, inv_scale, found_inf.data_ptr(),
                              _ptr(grad_norm), workspace.data_ptr(), workspace.numel(),
                              N.current_stream_ptr(params.device)), "ep_adamw_step")
    return found_inf


# --------------------------------------------------------------------------------------------
# autograd glue (module-level drop-in path)
# --------------------------------------------------------------------------------------------
class _EPPoolProject(torch.autograd.Function):
    """pooled = EfficientProbing(x) with gradients for cls_token and v.weight (x is frozen in
    the probing protocol; a gradient w.r.t. x is not provided)."""

    @staticmethod
    def forward(ctx, x, cls_token, v_weight, scale, per_image):
        P, S, ML = pool_forward(x, cls_token, scale, per_image_queries=per_image)
        y = project_forward(P, v_weight)
        ctx.save_for_backward(x, P, S, ML, y, v_weight)
        ctx.scale = scale
        ctx.per_image = per_image
        ctx.cls_shape = cls_token.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, P, S, ML, y, v_weight = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise RuntimeError("EfficientProbing (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        need_cls = ctx.needs_input_grad[1]
        dy = dy.contiguous()
        dP, dWv = project_backward(dy, y, P, v_weight, ML, need_dP=need_cls,
                                   need_dWv=ctx.needs_input_grad[2])
        dcls = None
        if need_cls and ctx.per_image:        # cls=... (reference poolings/ep.py:32-33): one gradient per image
            dcls = pool_backward_per_image(x, S, ML, dP, ctx.scale).reshape(ctx.cls_shape)
        elif need_cls:
            dcls = pool_backward(x, S, ML, dP, ctx.scale).reshape(ctx.cls_shape)
        return None, dcls, dWv, None, None


def ep_pool_project(x, cls_token, v_weight, scale, per_image=False):
    return _EPPoolProject.apply(x, cls_token, v_weight, scale, per_image)


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, running_mean, running_var, nbt, eps, momentum):
        z, rstd = bn_forward_train(y, running_mean, running_var, nbt, eps, momentum)
        ctx.save_for_backward(z, rstd)
        return z

    @staticmethod
    def backward(ctx, dz):
        z, rstd = ctx.saved_tensors
        return bn_backward(dz, z, rstd), None, None, None, None, None


def batch_norm_train(y, running_mean, running_var, nbt, eps=BN_EPS, momentum=BN_MOMENTUM):
    return _BatchNormTrain.apply(y, running_mean, running_var, nbt, eps, momentum)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, bias):
        ctx.save_for_backward(z, weight)
        ctx.has_bias = bias is not None
        return linear_forward(z, weight, bias)

    @staticmethod
    def backward(ctx, dlogits):
        z, weight = ctx.saved_tensors
        dz, dW, db = linear_backward(dlogits, z, weight, need_dz=ctx.needs_input_grad[0])
        return dz, dW, (db if ctx.has_bias else None)


def linear(z, weight, bias):
    return _Linear.apply(z, weight, bias)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        loss_rows, dl, stats = cross_entropy(logits, targets, 1.0, need_grad=True)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, gloss, _gstats):
        (dl,) = ctx.saved_tensors
        return dl * gloss, None


def cross_entropy_loss(logits, targets):
    """(loss, stats) with stats = [mean loss, #top1, #top5, #non-finite rows]."""
    return _CrossEntropy.apply(logits, targets)


# --------------------------------------------------------------------------------------------
# CoCa attentional pooler (reference poolings/coca_pytorch.py:250-343) on the EP token pass
# --------------------------------------------------------------------------------------------
def coca_dims(B, Nn, D, heads, dim_head, num_img_queries, C_=0):
    return N.EPCocaDims(B=B, N=Nn, D=D, H=heads, dh=dim_head, M=num_img_queries, C=C_)


def _coca_params_struct(gamma, beta, img_queries, to_q, to_kv, to_out):
    return N.EPCocaParams(gamma=gamma.data_ptr(), beta=_ptr(beta), img_queries=img_queries.data_ptr(),
                          to_q=to_q.data_ptr(), to_kv=to_kv.data_ptr(), to_out=to_out.data_ptr())


class _CocaPool(torch.autograd.Function):
    """out[:, 0] of the CoCa CrossAttention with gradients for its five parameter tensors (the tokens are
    frozen in the probing protocol; no gradient w.r.t. x)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, img_queries, to_q, to_kv, to_out, heads, dim_head, ln_eps):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in ((gamma, "norm.gamma"), (img_queries, "img_queries"), (to_q, "to_q.weight"),
                                          (to_kv, "to_kv.weight"), (to_out, "to_out.weight"))]
        beta_c = _f32c(beta, "norm.beta") if beta is not None else None
        dims = coca_dims(B, Nn, D, heads, dim_head, img_queries.shape[0])
        nbytes = lib.ep_coca_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_coca_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        ps = _coca_params_struct(tens[0], beta_c, *tens[1:])
        N.check(lib.ep_coca_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, C.byref(ps),
                                         float(ln_eps), y.data_ptr(), ws.data_ptr(), nbytes,
                                         N.current_stream_ptr(xv.device)), "ep_coca_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.beta = beta_c
        ctx.dims = dims
        ctx.bstride = bstride
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("CoCa pooler (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, gamma, imgq, to_q, to_kv, to_out = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in (gamma, imgq, to_q, to_kv, to_out)]
        ps = _coca_params_struct(gamma, ctx.beta, imgq, to_q, to_kv, to_out)
        gs = _coca_params_struct(grads[0], None, *grads[1:])
        N.check(lib.ep_coca_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0,
                                          C.byref(ps), dy.data_ptr(), C.byref(gs), 0, ws.data_ptr(), ws.numel(),
                                          N.current_stream_ptr(xv.device)), "ep_coca_pool_backward")
        return (None, grads[0], None, grads[1], grads[2], grads[3], grads[4], None, None, None)


def coca_pool(x, gamma, beta, img_queries, to_q, to_kv, to_out, heads, dim_head, ln_eps=1e-5):
    return _CocaPool.apply(x, gamma, beta, img_queries, to_q, to_kv, to_out, heads, dim_head, ln_eps)


def coca_attention(x, gamma, beta, img_queries, to_q, to_kv, to_out, heads, dim_head, ln_eps=1e-5):
    """softmax attention of image query 0 over the tokens, (B, heads, N)."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    B, Nn, D = xv.shape
    dims = coca_dims(B, Nn, D, heads, dim_head, img_queries.shape[0])
    nbytes = lib.ep_coca_pool_workspace_bytes(C.byref(dims))
    ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
    y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    ps = _coca_params_struct(_f32c(gamma, "gamma"), beta, _f32c(img_queries, "img_queries"), _f32c(to_q, "to_q"),
                             _f32c(to_kv, "to_kv"), _f32c(to_out, "to_out"))
    st = N.current_stream_ptr(xv.device)
    N.check(lib.ep_coca_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, C.byref(ps),
                                     float(ln_eps), y.data_ptr(), ws.data_ptr(), nbytes, st), "ep_coca_pool_forward")
    A = torch.empty((B, heads, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_coca_attention(C.byref(dims), ws.data_ptr(), A.data_ptr(), st), "ep_coca_attention")
    return A


# --------------------------------------------------------------------------------------------
# AbMILP head (reference poolings/abmilp.py:11-75 + models_vit.py:43-97): matrix-core bound
# --------------------------------------------------------------------------------------------
def _abmilp_params_struct(ts):
    return N.EPAbmilpParams(*[t.data_ptr() for t in ts])


def _contiguous_tokens(x):
    """AbMILP contracts over all B*N token rows at once, so the tokens must be one dense (B*N, D) matrix;
    a strided view (e.g. ``feat[:, 1:]``) is compacted once -- noise next to ~4 GFLOP per image."""
    N.require_gpu_tensor(x, "tokens")
    if x.dim() != 3:
        raise ValueError(f"tokens must be (B, N, D), got {tuple(x.shape)}")
    x = x.float() if x.dtype != torch.float32 else x
    return x if (x.is_contiguous() and x.data_ptr() % 16 == 0) else x.contiguous()


class _AbmilpPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, qkv, proj_w, proj_b, w1, b1, w2, b2):
        lib = N.load()
        xv = _contiguous_tokens(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in ((qkv, "qkv"), (proj_w, "proj.weight"), (proj_b, "proj.bias"), (w1, "w1"),
                                          (b1, "b1"), (w2, "w2"), (b2, "b2"))]
        dims = N.EPAbmilpDims(B=B, N=Nn, D=D, C=0)
        nbytes = lib.ep_abmilp_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_abmilp_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        amap = torch.empty((B, Nn), device=xv.device, dtype=torch.float32)
        ps = _abmilp_params_struct(tens)
        N.check(lib.ep_abmilp_pool_forward(C.byref(dims), xv.data_ptr(), N.EP_DTYPE_F32, Nn * D, C.byref(ps),
                                           out.data_ptr(), amap.data_ptr(), ws.data_ptr(), nbytes,
                                           N.current_stream_ptr(xv.device)), "ep_abmilp_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims = dims
        ctx.mark_non_differentiable(amap)
        return out, amap

    @staticmethod
    def backward(ctx, dout, _damap):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("AbMILP head (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dout = _f32c(dout, "dout")
        grads = [torch.empty_like(t) for t in tens]
        d = ctx.dims
        N.check(lib.ep_abmilp_pool_backward(C.byref(d), xv.data_ptr(), N.EP_DTYPE_F32, d.N * d.D,
                                            C.byref(_abmilp_params_struct(tens)), dout.data_ptr(),
                                            C.byref(_abmilp_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                            N.current_stream_ptr(xv.device)), "ep_abmilp_pool_backward")
        return (None, *grads)


def abmilp_pool(x, qkv, proj_w, proj_b, w1, b1, w2, b2):
    """(out (B, D), attention map (B, N)) of the AbMILP head."""
    return _AbmilpPool.apply(x, qkv, proj_w, proj_b, w1, b1, w2, b2)


# --------------------------------------------------------------------------------------------
# SigLIP attention-pool head (reference poolings/clip/attention_pool.py:13-140) on the EP token pass
# --------------------------------------------------------------------------------------------
SIGLIP_TENSORS = ("latent", "q.weight", "q.bias", "kv.weight", "kv.bias", "proj.weight", "proj.bias",
                  "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def _siglip_params_struct(ts):
    return N.EPSiglipParams(*[t.data_ptr() for t in ts])


class _SiglipPool(torch.autograd.Function):
    """AttentionPoolLatent(x) with gradients for its eleven parameter tensors (no gradient w.r.t. the tokens)."""

    @staticmethod
    def forward(ctx, x, heads, hidden, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, SIGLIP_TENSORS)]
        dims = N.EPSiglipDims(B=B, N=Nn, D=D, H=heads, hidden=hidden, C=0)
        nbytes = lib.ep_siglip_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_siglip_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_siglip_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0,
                                           C.byref(_siglip_params_struct(tens)), out.data_ptr(), ws.data_ptr(), nbytes,
                                           N.current_stream_ptr(xv.device)), "ep_siglip_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("SigLIP attention pool (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dout = _f32c(dout, "dout")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_siglip_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0,
                                            C.byref(_siglip_params_struct(tens)), dout.data_ptr(),
                                            C.byref(_siglip_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                            N.current_stream_ptr(xv.device)), "ep_siglip_pool_backward")
        return (None, None, None, *grads)


def siglip_pool(x, heads, hidden, *tens):
    return _SiglipPool.apply(x, heads, hidden, *tens)


def siglip_attention(x, heads, hidden, *tens):
    """softmax attention of the latent query over the tokens, (B, heads, N)."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    B, Nn, D = xv.shape
    tens = [_f32c(t, n) for t, n in zip(tens, SIGLIP_TENSORS)]
    dims = N.EPSiglipDims(B=B, N=Nn, D=D, H=heads, hidden=hidden, C=0)
    nbytes = lib.ep_siglip_pool_workspace_bytes(C.byref(dims))
    ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
    out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    st = N.current_stream_ptr(xv.device)
    N.check(lib.ep_siglip_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0,
                                       C.byref(_siglip_params_struct(tens)), out.data_ptr(), ws.data_ptr(), nbytes, st),
            "ep_siglip_pool_forward")
    A = torch.empty((B, heads, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_siglip_attention(C.byref(dims), ws.data_ptr(), A.data_ptr(), st), "ep_siglip_attention")
    return A


# --------------------------------------------------------------------------------------------
# LayerNorm-of-tokens mode of the token passes (heads that layer-norm every token before k / v)
# --------------------------------------------------------------------------------------------
def token_stats(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """(B, N, 2) per-token {mean, rstd} of a LayerNorm over D (biased variance, eps inside the root)."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    B, Nn, D = xv.shape
    out = torch.empty((B, Nn, 2), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_token_stats(xv.data_ptr(), token_dtype_code(xv), bstride, B, Nn, D, float(eps), out.data_ptr(),
                               N.current_stream_ptr(xv.device)), "ep_token_stats")
    return out


def token_xhat_mean(x: torch.Tensor, stats: torch.Tensor, image_index=None) -> torch.Tensor:
    """(B, D) mean over the N tokens of the normalised rows (x - mean) * rstd, given ``stats`` = token_stats(x, eps): the CLIP
    head's mean-row query input (reference poolings/clip/attention_pool2d.py:153-155).  A function of the frozen tokens only:
    a resident store keeps it as a table (``ResidentTokenStore.table("xhat_mean", eps)``)."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    stats = _f32c(stats, "token_stats")
    out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_token_xhat_mean(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, stats.data_ptr(), B, Nn, D, out.data_ptr(),
                                   N.current_stream_ptr(xv.device)), "ep_token_xhat_mean")
    return out


def pool_forward_ln(x, cls_token, scale, stats, image_index=None):
    """EP pooling of the NORMALISED tokens xhat = (x - mean) * rstd without materialising them.
    Returns P (B,Q,D) = softmax_n(S) xhat, S (B,Q,N) = (cls*scale) . xhat, ML (B,Q,4)."""
    lib = N.load()
    x, bstride = as_token_view(x)
    _, Nn, D = x.shape
    iptr, B = _index_arg(image_index, x)
    cls = _f32c(cls_token, "cls_token")
    stats = _f32c(stats, "token_stats")
    Q = cls.shape[-2]
    P = torch.empty((B, Q, D), device=x.device, dtype=torch.float32)
    S = torch.empty((B, Q, Nn), device=x.device, dtype=torch.float32)
    ML = torch.empty((B, Q, 4), device=x.device, dtype=torch.float32)
    N.check(lib.ep_pool_forward_ln(x.data_ptr(), token_dtype_code(x), bstride, iptr, B, Nn, D, cls.data_ptr(), 0, Q,
                                   float(scale), stats.data_ptr(), P.data_ptr(), S.data_ptr(), ML.data_ptr(), 0, 0,
                                   N.current_stream_ptr(x.device)), "ep_pool_forward_ln")
    return P, S, ML


def pool_backward_ln(x, S, ML, dP, scale, stats, image_index=None):
    lib = N.load()
    x, bstride = as_token_view(x)
    _, Nn, D = x.shape
    iptr, B = _index_arg(image_index, x)
    Q = S.shape[1]
    dP = _f32c(dP, "dP")
    stats = _f32c(stats, "token_stats")
    dcls = torch.empty((Q, D), device=x.device, dtype=torch.float32)
    nbytes = lib.ep_pool_workspace_bytes(B, Nn, D, Q)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    N.check(lib.ep_pool_backward_ln(x.data_ptr(), token_dtype_code(x), bstride, iptr, B, Nn, D, Q, float(scale),
                                    stats.data_ptr(), S.data_ptr(), ML.data_ptr(), dP.data_ptr(), dcls.data_ptr(), 0,
                                    ws.data_ptr(), nbytes, N.current_stream_ptr(x.device)), "ep_pool_backward_ln")
    return dcls


# --------------------------------------------------------------------------------------------
# CAE attentive block (reference poolings/cae_att.py:79-108) on the LayerNorm-of-tokens token passes
# --------------------------------------------------------------------------------------------
CAE_TENSORS = ("query_token", "norm1_q.weight", "norm1_q.bias", "norm1_k.weight", "norm1_k.bias", "norm1_v.weight",
               "norm1_v.bias", "norm2_cross.weight", "norm2_cross.bias", "cross_attn.q.weight", "cross_attn.k.weight",
               "cross_attn.v.weight", "cross_attn.proj.weight", "cross_attn.proj.bias")
CAE_LN_EPS = 1e-5          # nn.LayerNorm default (cae_att.py:82 norm_layer=nn.LayerNorm)


def _cae_params_struct(ts):
    return N.EPCaeParams(*[t.data_ptr() for t in ts])


class _CaePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, CAE_TENSORS)]
        dims = N.EPCaeDims(B=B, N=Nn, D=D, H=heads, C=0)
        nbytes = lib.ep_cae_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_cae_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_cae_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, CAE_LN_EPS,
                                        C.byref(_cae_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes,
                                        N.current_stream_ptr(xv.device)), "ep_cae_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("CAE attentive block (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_cae_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0, 0, CAE_LN_EPS,
                                         C.byref(_cae_params_struct(tens)), dy.data_ptr(), C.byref(_cae_params_struct(grads)),
                                         0, ws.data_ptr(), ws.numel(), N.current_stream_ptr(xv.device)),
                "ep_cae_pool_backward")
        return (None, None, *grads)


def cae_pool(x, heads, *tens):
    return _CaePool.apply(x, heads, *tens)


# --------------------------------------------------------------------------------------------
# V-JEPA attentive pooler (reference poolings/jepa/attentive_pooler.py:21-104) on the LayerNorm-of-tokens passes
# --------------------------------------------------------------------------------------------
JEPA_TENSORS = ("query_tokens", "norm1.weight", "norm1.bias", "xattn.q.weight", "xattn.q.bias", "xattn.kv.weight",
                "xattn.kv.bias", "xattn.proj.weight", "xattn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight",
                "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
JEPA_LN_EPS = 1e-5


def _jepa_params_struct(ts):
    return N.EPJepaParams(*[t.data_ptr() for t in ts])


class _JepaPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, hidden, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, JEPA_TENSORS)]
        dims = N.EPJepaDims(B=B, N=Nn, D=D, H=heads, hidden=hidden, C=0)
        nbytes = lib.ep_jepa_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_jepa_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_jepa_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, JEPA_LN_EPS,
                                         C.byref(_jepa_params_struct(tens)), out.data_ptr(), ws.data_ptr(), nbytes,
                                         N.current_stream_ptr(xv.device)), "ep_jepa_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("JEPA attentive pooler (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dout = _f32c(dout, "dout")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_jepa_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0, 0,
                                          C.byref(_jepa_params_struct(tens)), dout.data_ptr(),
                                          C.byref(_jepa_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                          N.current_stream_ptr(xv.device)), "ep_jepa_pool_backward")
        return (None, None, None, *grads)


def jepa_pool(x, heads, hidden, *tens):
    return _JepaPool.apply(x, heads, hidden, *tens)


# --------------------------------------------------------------------------------------------
# AIM attention-pooling head (reference poolings/aim.py:337-392) on the plain EP token passes
# --------------------------------------------------------------------------------------------
AIM_TENSORS = ("cls_token", "k.weight", "v.weight")
AIM_BN_EPS = 1e-6          # aim.py:357 BatchNorm1d(dim, affine=False, eps=1e-6)


def _aim_params_struct(ts):
    return N.EPAimParams(*[t.data_ptr() for t in ts])


def channel_stats(x: torch.Tensor, image_index=None) -> torch.Tensor:
    """Per-image column statistics {mean over the N tokens, sum of squared deviations} -> (B, 2, D) fp32.
    They depend on the frozen tokens only: compute them once for a resident token store and hand the table to
    ``AimHeadEngine.train_step(..., image_stats=...)``."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    out = torch.empty((B, 2, D), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_channel_stats(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, B, Nn, D, out.data_ptr(),
                                 N.current_stream_ptr(xv.device)), "ep_channel_stats")
    return out


class _AimPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, training, eps, momentum, running_mean, running_var, nbt, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, AIM_TENSORS)]
        dims = N.EPAimDims(B=B, N=Nn, D=D, H=heads, C=0)
        nbytes = lib.ep_aim_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_aim_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_aim_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, int(training),
                                        float(eps), float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                        C.byref(_aim_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes,
                                        N.current_stream_ptr(xv.device)), "ep_aim_pool_forward")
        ctx.save_for_backward(xv, ws, y, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("AIM attention pooling (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, y, *tens = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_aim_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0,
                                         C.byref(_aim_params_struct(tens)), y.data_ptr(), dy.data_ptr(),
                                         C.byref(_aim_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                         N.current_stream_ptr(xv.device)), "ep_aim_pool_backward")
        return (None,) * 8 + tuple(grads)


def aim_pool(x, heads, training, eps, momentum, running_mean, running_var, nbt, *tens):
    return _AimPool.apply(x, heads, training, eps, momentum, running_mean, running_var, nbt, *tens)


def aim_attention(x, heads, training, eps, running_mean, running_var, *tens):
    """Attention weights (B, H, N) of the AIM head (no statistics update)."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    B, Nn, D = xv.shape
    tens = [_f32c(t.detach(), n) for t, n in zip(tens, AIM_TENSORS)]
    dims = N.EPAimDims(B=B, N=Nn, D=D, H=heads, C=0)
    nbytes = lib.ep_aim_pool_workspace_bytes(C.byref(dims))
    ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
    y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    rm = running_mean.clone() if running_mean is not None else None      # batch statistics without side effects
    rv = running_var.clone() if running_var is not None else None
    N.check(lib.ep_aim_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, int(training),
                                    float(eps), 0.0, _ptr(rm), _ptr(rv), 0, C.byref(_aim_params_struct(tens)),
                                    y.data_ptr(), ws.data_ptr(), nbytes, N.current_stream_ptr(xv.device)),
            "ep_aim_pool_forward")
    A = torch.empty((B, heads, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_aim_attention(C.byref(dims), ws.data_ptr(), A.data_ptr(), N.current_stream_ptr(xv.device)),
            "ep_aim_attention")
    return A


# --------------------------------------------------------------------------------------------
# SimPool heads (reference poolings/simpool.py) on the per-image-query token passes
# --------------------------------------------------------------------------------------------
SIMPOOL_LN_EPS = 1e-6      # simpool.py:12 / :100  nn.LayerNorm(dim, eps=1e-6)


def _simpool_params_struct(ts):
    ts = list(ts) + [None] * (4 - len(ts))
    return N.EPSimpoolParams(*[_ptr(t) for t in ts])


class _SimPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, linears, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        names = ("norm_patches.weight", "norm_patches.bias", "wq.weight", "wk.weight")
        tens = [_f32c(t, n) for t, n in zip(tens, names)]
        dims = N.EPSimpoolDims(B=B, N=Nn, D=D, H=heads, C=0, linears=int(linears))
        nbytes = lib.ep_simpool_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_simpool_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_simpool_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, 0, SIMPOOL_LN_EPS,
                                            C.byref(_simpool_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes,
                                            N.current_stream_ptr(xv.device)), "ep_simpool_pool_forward")
        ctx.save_for_backward(xv, ws, y, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("SimPool (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, y, *tens = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_simpool_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0, 0, 0,
                                             SIMPOOL_LN_EPS, C.byref(_simpool_params_struct(tens)), y.data_ptr(), dy.data_ptr(),
                                             C.byref(_simpool_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                             N.current_stream_ptr(xv.device)), "ep_simpool_pool_backward")
        return (None, None, None, *grads)


def simpool_pool(x, heads, linears, *tens):
    return _SimPool.apply(x, heads, linears, *tens)


def simpool_attention(x, heads, linears, *tens):
    """(pooled (B, D), attention (B, H, N)) of a SimPool head (reference simpool.py return_attn=True)."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    B, Nn, D = xv.shape
    tens = [_f32c(t.detach(), "tensor") for t in tens]
    dims = N.EPSimpoolDims(B=B, N=Nn, D=D, H=heads, C=0, linears=int(linears))
    nbytes = lib.ep_simpool_pool_workspace_bytes(C.byref(dims))
    if nbytes == 0:
        raise RuntimeError(f"ep_simpool_pool_workspace_bytes: {N.last_error()}")
    ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
    y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    st = N.current_stream_ptr(xv.device)
    N.check(lib.ep_simpool_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, 0, SIMPOOL_LN_EPS,
                                        C.byref(_simpool_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes, st),
            "ep_simpool_pool_forward")
    A = torch.empty((B, heads, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_simpool_attention(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, ws.data_ptr(),
                                     A.data_ptr(), st), "ep_simpool_attention")
    return y, A


def imgq_pool_forward(x, u, heads, token_stats=None, pool_ln=False, image_index=None):
    """Per-image-query token pass over channel slices (csrc/ep_pool_imgq.hip): u (B, D) -> (P (B, D), ML (B, H, 2))."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    u = _f32c(u, "u")
    P = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    ML = torch.empty((B, heads, 2), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_imgq_pool_forward(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, B, Nn, D, heads, u.data_ptr(),
                                     _ptr(token_stats), int(pool_ln), P.data_ptr(), ML.data_ptr(),
                                     N.current_stream_ptr(xv.device)), "ep_imgq_pool_forward")
    return P, ML


def imgq_pool_backward(x, u, heads, P, ML, dP, token_stats=None, pool_ln=False, image_index=None):
    """d u (B, D) per image of the per-image-query token pass."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    u, dP = _f32c(u, "u"), _f32c(dP, "dP")
    du = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_imgq_pool_backward(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, B, Nn, D, heads, u.data_ptr(),
                                      _ptr(token_stats), int(pool_ln), P.data_ptr(), ML.data_ptr(), dP.data_ptr(),
                                      du.data_ptr(), N.current_stream_ptr(xv.device)), "ep_imgq_pool_backward")
    return du


def rowq_pool_forward(x, u, token_stats=None, score_bias=None, image_index=None):
    """Full-width per-image-query token pass (csrc/ep_pool_imgq.hip, ep_imgqf_kernel): u (B, Q <= 4, D) ->
    (P (B, Q, D), S (B, Q, N), ML (B, Q, 4))."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    u = _f32c(u, "u")
    Q = u.shape[1]
    P = torch.empty((B, Q, D), device=xv.device, dtype=torch.float32)
    S = torch.empty((B, Q, Nn), device=xv.device, dtype=torch.float32)
    ML = torch.empty((B, Q, 4), device=xv.device, dtype=torch.float32)
    sb = _f32c(score_bias, "score_bias") if score_bias is not None else None
    N.check(lib.ep_rowq_pool_forward(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, B, Nn, D, Q, u.data_ptr(), _ptr(token_stats),
                                     _ptr(sb), P.data_ptr(), S.data_ptr(), ML.data_ptr(), N.current_stream_ptr(xv.device)),
            "ep_rowq_pool_forward")
    return P, S, ML


def rowq_pool_backward(x, S, ML, dP, token_stats=None, dA_bias=None, want_dS=False, image_index=None):
    """(du (B, Q, D) per image, dS (B, Q, N) | None) of the full-width per-image-query token pass; ML[..., 2] = delta."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    dP = _f32c(dP, "dP")
    Q = dP.shape[1]
    du = torch.empty((B, Q, D), device=xv.device, dtype=torch.float32)
    dS = torch.empty((B, Q, Nn), device=xv.device, dtype=torch.float32) if want_dS else None
    db = _f32c(dA_bias, "dA_bias") if dA_bias is not None else None
    N.check(lib.ep_rowq_pool_backward(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, B, Nn, D, Q, _ptr(token_stats), S.data_ptr(),
                                      ML.data_ptr(), dP.data_ptr(), _ptr(db), _ptr(dS), du.data_ptr(),
                                      N.current_stream_ptr(xv.device)), "ep_rowq_pool_backward")
    return du, dS


# --------------------------------------------------------------------------------------------
# CaiT class-attention pooling (reference poolings/other_pool.py:390-507) on the LayerNorm-of-tokens passes
# --------------------------------------------------------------------------------------------
CAIT_TENSORS = ("cls_token", "gamma_1", "gamma_2", "norm1.weight", "norm1.bias", "attn.q.weight", "attn.q.bias", "attn.k.weight",
                "attn.k.bias", "attn.v.weight", "attn.v.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
                "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "norm.weight", "norm.bias")
CAIT_LN_EPS = 1e-6         # other_pool.py:395 norm_layer = partial(nn.LayerNorm, eps=1e-6)
CAIT_FINAL_EPS = 1e-5      # other_pool.py:415 nn.LayerNorm(embed_dim)


def _cait_params_struct(ts):
    return N.EPCaitParams(*[t.data_ptr() for t in ts])


def cait_dims(B, Nn, D, heads, hidden, C_=0):
    return N.EPCaitDims(B=B, N=Nn, D=D, H=heads, hidden=hidden, C=C_, ln_eps=CAIT_LN_EPS, final_eps=CAIT_FINAL_EPS)


class _CaitPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, hidden, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, CAIT_TENSORS)]
        dims = cait_dims(B, Nn, D, heads, hidden)
        nbytes = lib.ep_cait_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_cait_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_cait_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0,
                                         C.byref(_cait_params_struct(tens)), out.data_ptr(), ws.data_ptr(), nbytes,
                                         N.current_stream_ptr(xv.device)), "ep_cait_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("CaiT class-attention pooling (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dout = _f32c(dout, "dout")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_cait_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0, 0,
                                          C.byref(_cait_params_struct(tens)), dout.data_ptr(),
                                          C.byref(_cait_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                          N.current_stream_ptr(xv.device)), "ep_cait_pool_backward")
        return (None, None, None, *grads)


def cait_pool(x, heads, hidden, *tens):
    return _CaitPool.apply(x, heads, hidden, *tens)


# --------------------------------------------------------------------------------------------
# CLIP attention pooling (reference poolings/clip/attention_pool2d.py:100-169)
# --------------------------------------------------------------------------------------------
CLIP_TENSORS = ("pos_embed", "qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "norm.weight", "norm.bias")
CLIP_LN_EPS = 1e-6         # attention_pool2d.py:138 nn.LayerNorm(in_features, eps=1e-6)


def _clip_params_struct(ts):
    return N.EPClipParams(*[t.data_ptr() for t in ts])


def clip_dims(B, Nn, D, heads, C_=0):
    return N.EPClipDims(B=B, N=Nn, D=D, H=heads, C=C_, ln_eps=CLIP_LN_EPS)


class _ClipPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, CLIP_TENSORS)]
        dims = clip_dims(B, Nn, D, heads)
        nbytes = lib.ep_clip_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_clip_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_clip_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0,
                                         C.byref(_clip_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes,
                                         N.current_stream_ptr(xv.device)), "ep_clip_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("CLIP attention pooling (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_clip_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0, 0,
                                          C.byref(_clip_params_struct(tens)), dy.data_ptr(),
                                          C.byref(_clip_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                          N.current_stream_ptr(xv.device)), "ep_clip_pool_backward")
        return (None, None, *grads)


def clip_pool(x, heads, *tens):
    return _ClipPool.apply(x, heads, *tens)


def clip_attention(x, heads, *tens):
    """(pooled (B, D), attention of the mean-row query over the patch rows (B, H, N))."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    B, Nn, D = xv.shape
    tens = [_f32c(t.detach(), n) for t, n in zip(tens, CLIP_TENSORS)]
    dims = clip_dims(B, Nn, D, heads)
    nbytes = lib.ep_clip_pool_workspace_bytes(C.byref(dims))
    if nbytes == 0:
        raise RuntimeError(f"ep_clip_pool_workspace_bytes: {N.last_error()}")
    ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
    y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    st = N.current_stream_ptr(xv.device)
    N.check(lib.ep_clip_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0,
                                     C.byref(_clip_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes, st),
            "ep_clip_pool_forward")
    A = torch.empty((B, heads, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_clip_attention(C.byref(dims), ws.data_ptr(), A.data_ptr(), st), "ep_clip_attention")
    return y, A


# --------------------------------------------------------------------------------------------
# DOLG spatial attention pooling (reference poolings/dolg/dolg.py:11-62): matrix-core bound
# --------------------------------------------------------------------------------------------
DOLG_TENSORS = ("conv1.weight", "conv1.bias", "bn.weight", "bn.bias", "conv2.weight", "conv2.bias")


def _dolg_params_struct(ts):
    return N.EPDolgParams(*[t.data_ptr() for t in ts])


class _DolgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, training, eps, momentum, running_mean, running_var, nbt, *tens):
        lib = N.load()
        xv = _contiguous_tokens(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, DOLG_TENSORS)]
        dims = N.EPDolgDims(B=B, N=Nn, D=D, C=0)
        nbytes = lib.ep_dolg_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_dolg_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_dolg_pool_forward(C.byref(dims), xv.data_ptr(), N.EP_DTYPE_F32, Nn * D, int(training), float(eps),
                                         float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                         C.byref(_dolg_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes,
                                         N.current_stream_ptr(xv.device)), "ep_dolg_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims = dims
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("DOLG spatial attention (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in tens]
        d = ctx.dims
        N.check(lib.ep_dolg_pool_backward(C.byref(d), xv.data_ptr(), N.EP_DTYPE_F32, d.N * d.D, C.byref(_dolg_params_struct(tens)),
                                          dy.data_ptr(), C.byref(_dolg_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                          N.current_stream_ptr(xv.device)), "ep_dolg_pool_backward")
        return (None,) * 7 + tuple(grads)


def dolg_pool(x, training, eps, momentum, running_mean, running_var, nbt, *tens):
    return _DolgPool.apply(x, training, eps, momentum, running_mean, running_var, nbt, *tens)


def dolg_attention(x, training, eps, running_mean, running_var, *tens):
    """(pooled (B, D), softplus attention scores (B, N)) -- no statistics update."""
    lib = N.load()
    xv = _contiguous_tokens(x)
    B, Nn, D = xv.shape
    tens = [_f32c(t.detach(), n) for t, n in zip(tens, DOLG_TENSORS)]
    dims = N.EPDolgDims(B=B, N=Nn, D=D, C=0)
    nbytes = lib.ep_dolg_pool_workspace_bytes(C.byref(dims))
    if nbytes == 0:
        raise RuntimeError(f"ep_dolg_pool_workspace_bytes: {N.last_error()}")
    ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
    y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    rm = running_mean.clone() if running_mean is not None else None
    rv = running_var.clone() if running_var is not None else None
    st = N.current_stream_ptr(xv.device)
    N.check(lib.ep_dolg_pool_forward(C.byref(dims), xv.data_ptr(), N.EP_DTYPE_F32, Nn * D, int(training), float(eps), 0.0, _ptr(rm),
                                     _ptr(rv), 0, C.byref(_dolg_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes, st),
            "ep_dolg_pool_forward")
    att = torch.empty((B, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_dolg_attention(C.byref(dims), ws.data_ptr(), att.data_ptr(), st), "ep_dolg_attention")
    return y, att


# --------------------------------------------------------------------------------------------
# CBAM pooling (reference poolings/cbam.py:104-139): streaming passes over the tokens
# --------------------------------------------------------------------------------------------
CBAM_TENSORS = ("channel.fc1.weight", "channel.fc2.weight", "spatial.conv.conv.weight", "spatial.conv.bn.weight",
                "spatial.conv.bn.bias")


def _cbam_params_struct(ts):
    return N.EPCbamParams(*[t.data_ptr() for t in ts])


def cbam_channel_table(x: torch.Tensor, image_index=None) -> torch.Tensor:
    """Per-image {mean_n x, max_n x, mean_n relu(x)} per channel -> (B, 3, D) fp32: compute once for a resident token store and
    hand it to ``CbamHeadEngine.train_step(..., image_stats=...)``."""
    lib = N.load()
    xv, bstride = as_token_view(x)
    iptr, B = _index_arg(image_index, xv)
    _, Nn, D = xv.shape
    out = torch.empty((B, 3, D), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_cbam_channel_table(xv.data_ptr(), token_dtype_code(xv), bstride, iptr, B, Nn, D, out.data_ptr(),
                                      N.current_stream_ptr(xv.device)), "ep_cbam_channel_table")
    return out


class _CbamPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rd, ks, training, eps, momentum, running_mean, running_var, nbt, *tens):
        lib = N.load()
        xv, bstride = as_token_view(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, CBAM_TENSORS)]
        dims = N.EPCbamDims(B=B, N=Nn, D=D, C=0, rd=rd, ks=ks)
        nbytes = lib.ep_cbam_pool_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError(f"ep_cbam_pool_workspace_bytes: {N.last_error()}")
        ws = torch.empty(nbytes, device=xv.device, dtype=torch.uint8)
        y = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_cbam_pool_forward(C.byref(dims), xv.data_ptr(), token_dtype_code(xv), bstride, 0, 0, int(training), float(eps),
                                         float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                         C.byref(_cbam_params_struct(tens)), y.data_ptr(), ws.data_ptr(), nbytes,
                                         N.current_stream_ptr(xv.device)), "ep_cbam_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims, ctx.bstride = dims, bstride
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("CBAM pooling (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        grads = [torch.empty_like(t) for t in tens]
        N.check(lib.ep_cbam_pool_backward(C.byref(ctx.dims), xv.data_ptr(), token_dtype_code(xv), ctx.bstride, 0,
                                          C.byref(_cbam_params_struct(tens)), dy.data_ptr(), C.byref(_cbam_params_struct(grads)), 0,
                                          ws.data_ptr(), ws.numel(), N.current_stream_ptr(xv.device)), "ep_cbam_pool_backward")
        return (None,) * 9 + tuple(grads)


def cbam_pool(x, rd, ks, training, eps, momentum, running_mean, running_var, nbt, *tens):
    return _CbamPool.apply(x, rd, ks, training, eps, momentum, running_mean, running_var, nbt, *tens)


# --------------------------------------------------------------------------------------------
# DINOv2-block pooling (reference poolings/other_pool.py:299-318 + dinov2_layers/block.py:43-113): matrix-core bound
# --------------------------------------------------------------------------------------------
DINOVIT_TENSORS = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.proj.weight", "attn.proj.bias", "norm2.weight",
                   "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
DINOVIT_LN_EPS = 1e-5


def _dinovit_params_struct(ts):
    return N.EPDinovitParams(*[t.data_ptr() for t in ts])


def dinovit_dims(B, Nn, D, H, hidden, n_classes=0, eps=DINOVIT_LN_EPS):
    return N.EPDinovitDims(B=B, N=Nn, D=D, H=H, hidden=hidden, C=n_classes, ln_eps=eps)


def _dinovit_ws(lib, dims, device):
    nbytes = lib.ep_dinovit_pool_workspace_bytes(C.byref(dims))
    if nbytes == 0:
        raise RuntimeError(f"ep_dinovit_pool_workspace_bytes: {N.last_error()}")
    return torch.empty(nbytes, device=device, dtype=torch.uint8)


class _DinovitPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, eps, *tens):
        lib = N.load()
        xv = _contiguous_tokens(x)
        B, Nn, D = xv.shape
        tens = [_f32c(t, n) for t, n in zip(tens, DINOVIT_TENSORS)]
        dims = dinovit_dims(B, Nn, D, H, tens[7].shape[0], eps=eps)
        ws = _dinovit_ws(lib, dims, xv.device)
        out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
        N.check(lib.ep_dinovit_pool_forward(C.byref(dims), xv.data_ptr(), N.EP_DTYPE_F32, Nn * D, C.byref(_dinovit_params_struct(tens)),
                                            out.data_ptr(), ws.data_ptr(), ws.numel(), N.current_stream_ptr(xv.device)),
                "ep_dinovit_pool_forward")
        ctx.save_for_backward(xv, ws, *tens)
        ctx.dims = dims
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("DINOv2-block pooling (native): gradient w.r.t. the tokens is not implemented -- "
                               "the probe trains on a frozen encoder (detach the tokens)")
        lib = N.load()
        xv, ws, *tens = ctx.saved_tensors
        dout = _f32c(dout, "dout")
        grads = [torch.empty_like(t) for t in tens]
        d = ctx.dims
        N.check(lib.ep_dinovit_pool_backward(C.byref(d), xv.data_ptr(), N.EP_DTYPE_F32, d.N * d.D, C.byref(_dinovit_params_struct(tens)),
                                             dout.data_ptr(), C.byref(_dinovit_params_struct(grads)), 0, ws.data_ptr(), ws.numel(),
                                             N.current_stream_ptr(xv.device)), "ep_dinovit_pool_backward")
        return (None, None, None, *grads)


def dinovit_pool(x, H, eps, *tens):
    """mean over the tokens of one DINOv2 block: (B, D)."""
    return _DinovitPool.apply(x, H, eps, *tens)


def dinovit_attention(x, H, eps, *tens):
    """(pooled (B, D), attention weights (B, H, N, N)) of the block -- no autograd."""
    lib = N.load()
    xv = _contiguous_tokens(x)
    B, Nn, D = xv.shape
    tens = [_f32c(t.detach(), n) for t, n in zip(tens, DINOVIT_TENSORS)]
    dims = dinovit_dims(B, Nn, D, H, tens[7].shape[0], eps=eps)
    ws = _dinovit_ws(lib, dims, xv.device)
    out = torch.empty((B, D), device=xv.device, dtype=torch.float32)
    st = N.current_stream_ptr(xv.device)
    N.check(lib.ep_dinovit_pool_forward(C.byref(dims), xv.data_ptr(), N.EP_DTYPE_F32, Nn * D, C.byref(_dinovit_params_struct(tens)),
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), st), "ep_dinovit_pool_forward")
    A = torch.empty((B, H, Nn, Nn), device=xv.device, dtype=torch.float32)
    N.check(lib.ep_dinovit_attention(C.byref(dims), ws.data_ptr(), A.data_ptr(), st), "ep_dinovit_attention")
    return out, A
