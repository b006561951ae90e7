"""CPU oracle for the efficient-probing (EP) head hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm
(billpsomas/efficient-probing, mounted read-only at /root/reference while it was
written).  It exists so that the HIP path can be checked on a machine where the
reference does not exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``efficient_probing_amd``) never does.

PARITY PIN: every function here is checked against golden vectors produced by
importing the real reference modules (``poolings/ep.py``, ``util/lars.py``,
``util/lr_sched.py``) and stock ``torch`` ops (BatchNorm1d / Linear /
CrossEntropyLoss / GradScaler) in the build container -- see
``tests/golden/make_golden.py`` (the generating script) and
``tests/test_oracle_golden.py``.  The reference itself ships no forward /
gradient golden vectors, so those fixtures are the pin.

The forward keeps the reference's association (project every token with ``v``,
then pool) and the backward is what autograd produces for that graph; it does
NOT use the pool-then-project refactoring of the HIP kernels, so agreement
between the two is a real check of that refactoring.

All arithmetic is float32 unless a function says otherwise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# EP pooling: poolings/ep.py
# --------------------------------------------------------------------------- #
def ep_scale(dim: int, num_heads: int = 1, qk_scale: Optional[float] = None) -> float:
    """``self.scale = qk_scale or head_dim ** -0.5`` (poolings/ep.py:19-20)."""
    head_dim = dim // num_heads
    return qk_scale or head_dim ** -0.5


def _softmax_lastdim(s: np.ndarray) -> np.ndarray:
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m, dtype=F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def ep_forward(x: np.ndarray, cls_token: np.ndarray, v_weight: np.ndarray,
               num_queries: int, d_out: int = 1, num_heads: int = 1,
               qk_scale: Optional[float] = None, cls: Optional[np.ndarray] = None):
    """``EfficientProbing.forward`` (poolings/ep.py:28-47), num_heads == 1.

    x (B,N,C) ; cls_token (1,Q,C) ; v_weight (C//d_out, C).
    Returns ``(x_cls (B, C//d_out), cache)``; ``cache`` holds what the backward
    below needs (attn (B,Q,N), v (B,Q,N,Dq)).
    """
    if num_heads != 1:
        # ep.py:44 squeezes dim 1 of attn (B,H,Q,N); only H == 1 is meaningful and the
        # registry (probe_heads.py:75) never passes num_heads.
        raise NotImplementedError("reference EP is only well defined for num_heads == 1")
    x = np.asarray(x, dtype=F32)
    B, N, C = x.shape
    Q = num_queries
    c_prime = C // d_out                                           # ep.py:30
    scale = F32(ep_scale(C, num_heads, qk_scale))
    tok = cls if cls is not None else np.broadcast_to(cls_token, (B, Q, C))  # ep.py:32-35
    q = (np.asarray(tok, dtype=F32) * scale).astype(F32)           # ep.py:37,39  (B,Q,C)
    k = x                                                          # ep.py:38    (B,N,C)
    v = (x.reshape(B * N, C) @ v_weight.T.astype(F32)).astype(F32) # ep.py:40 Linear, no bias
    dq = c_prime // Q
    v = v.reshape(B, N, Q, dq).transpose(0, 2, 1, 3)               # (B,Q,N,Dq)
    attn = np.matmul(q, k.transpose(0, 2, 1)).astype(F32)          # ep.py:42   (B,Q,N)
    attn = _softmax_lastdim(attn)                                  # ep.py:43
    x_cls = np.matmul(attn[:, :, None, :], v).astype(F32)          # ep.py:44   (B,Q,1,Dq)
    out = x_cls.reshape(B, c_prime)                                # ep.py:45
    cache = dict(x=x, attn=attn, v=v, scale=scale, Q=Q, dq=dq, per_image=cls is not None)
    return out, cache


def ep_attention(x: np.ndarray, cls_token: np.ndarray, num_heads: int = 1,
                 qk_scale: Optional[float] = None) -> np.ndarray:
    """softmax((cls_token * C**-0.5) @ x^T) -- the attention maps as restated by
    tools/ep_attention_maps.py:52-58.  Returns (B,Q,N)."""
    x = np.asarray(x, dtype=F32)
    B, N, C = x.shape
    scale = F32(ep_scale(C, num_heads, qk_scale))
    q = (cls_token.astype(F32) * scale).astype(F32)                # (1,Q,C)
    s = np.matmul(q, x.transpose(0, 2, 1)).astype(F32)
    return _softmax_lastdim(s)


def ep_backward(dout: np.ndarray, cache: dict, v_weight: np.ndarray):
    """Autograd of ep_forward w.r.t. ``v.weight`` and ``cls_token`` (x is frozen:
    main_linprobe.py:393-400, so no dx).  Returns (dcls_token (1,Q,C), dv_weight); after a forward with per-image
    queries (``cls=``, ep.py:32-33) the first result is the gradient of those queries, (B,Q,C)."""
    x, attn, v, scale, Q, dq = (cache[k] for k in ("x", "attn", "v", "scale", "Q", "dq"))
    B, N, C = x.shape
    d = np.asarray(dout, dtype=F32).reshape(B, Q, 1, dq)
    # x_cls = attn[:, :, None, :] @ v
    dattn = np.matmul(d, v.transpose(0, 1, 3, 2)).astype(F32)[:, :, 0, :]     # (B,Q,N)
    dv = (attn[:, :, :, None] * d).astype(F32)                                # (B,Q,N,Dq)
    # v = Linear(x) reshaped/permuted
    dv_full = dv.transpose(0, 2, 1, 3).reshape(B * N, Q * dq)                 # (BN, C')
    dv_weight = (dv_full.T @ x.reshape(B * N, C)).astype(F32)                 # (C', C)
    # softmax
    inner = (attn * dattn).sum(axis=-1, keepdims=True, dtype=F32)
    ds = (attn * (dattn - inner)).astype(F32)                                 # (B,Q,N)
    # attn_logits = (cls_token * scale) @ x^T ; cls_token expanded over the batch
    dq_tok = np.matmul(ds, x).astype(F32)                                     # (B,Q,C)
    if cache.get("per_image"):                                                # cls= given: no expand, so no batch sum
        return (dq_tok * scale).astype(F32), dv_weight
    dcls = (dq_tok.sum(axis=0, dtype=F32) * scale).astype(F32)[None]          # (1,Q,C)
    return dcls, dv_weight


# --------------------------------------------------------------------------- #
# BatchNorm1d(affine=False, eps=1e-6): probe_heads.py:109-110
# --------------------------------------------------------------------------- #
BN_EPS = 1e-6
BN_MOMENTUM = 0.1


def bn_forward_train(y, running_mean, running_var, num_batches_tracked,
                     eps: float = BN_EPS, momentum: float = BN_MOMENTUM):
    """torch.nn.BatchNorm1d(width, affine=False, eps=1e-6) in train mode
    (probe_heads.py:109-110, used at :106).  Batch statistics are biased; the running
    variance is updated with the unbiased estimate.  Returns
    (z, new_rm, new_rv, new_nbt, cache)."""
    y = np.asarray(y, dtype=F32)
    B = y.shape[0]
    mu = y.mean(axis=0, dtype=F32)
    var = ((y - mu) ** 2).mean(axis=0, dtype=F32)
    rstd = (1.0 / np.sqrt(var + F32(eps))).astype(F32)
    z = ((y - mu) * rstd).astype(F32)
    unbiased = var * F32(B / max(B - 1, 1))
    new_rm = ((1 - momentum) * running_mean + momentum * mu).astype(F32)
    new_rv = ((1 - momentum) * running_var + momentum * unbiased).astype(F32)
    return z, new_rm, new_rv, int(num_batches_tracked) + 1, dict(z=z, rstd=rstd)


def bn_forward_eval(y, running_mean, running_var, eps: float = BN_EPS):
    return ((np.asarray(y, dtype=F32) - running_mean) / np.sqrt(running_var + F32(eps))).astype(F32)


def bn_backward_train(dz, cache):
    z, rstd = cache["z"], cache["rstd"]
    dz = np.asarray(dz, dtype=F32)
    m1 = dz.mean(axis=0, dtype=F32)
    m2 = (dz * z).mean(axis=0, dtype=F32)
    return (rstd * (dz - m1 - z * m2)).astype(F32)


# --------------------------------------------------------------------------- #
# classifier Linear (probe_heads.py:76), CrossEntropyLoss (main_linprobe.py:589),
# timm.utils.accuracy (engine_finetune.py:63)
# --------------------------------------------------------------------------- #
def linear_forward(z, weight, bias):
    return (np.asarray(z, dtype=F32) @ weight.T.astype(F32) + bias).astype(F32)


def linear_backward(dlogits, z, weight):
    dW = (dlogits.T @ z).astype(F32)
    db = dlogits.sum(axis=0, dtype=F32)
    dz = (dlogits @ weight).astype(F32)
    return dz, dW, db


def cross_entropy(logits, targets):
    """Mean CE over the batch; returns (loss, dlogits_of_mean_loss)."""
    logits = np.asarray(logits, dtype=F32)
    B = logits.shape[0]
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m, dtype=F32)
    se = e.sum(axis=1, keepdims=True, dtype=F32)
    logp = (logits - m - np.log(se)).astype(F32)
    loss = F32(-logp[np.arange(B), targets].mean(dtype=F32))
    p = (e / se).astype(F32)
    p[np.arange(B), targets] -= F32(1.0)
    return loss, (p / F32(B)).astype(F32)


def accuracy(logits, targets, topk=(1, 5)):
    """timm.utils.accuracy: percentage of rows whose target is within the top-k logits
    (torch.topk order; ties broken by lower index first, as a stable sort does)."""
    logits = np.asarray(logits)
    B, C = logits.shape
    maxk = min(max(topk), C)
    order = np.argsort(-logits, axis=1, kind="stable")[:, :maxk]
    hit = order == np.asarray(targets)[:, None]
    return [float(hit[:, :min(k, maxk)].sum() * 100.0 / B) for k in topk]


# --------------------------------------------------------------------------- #
# Schedules and optimizers
# --------------------------------------------------------------------------- #
def adjust_learning_rate(epoch: float, lr: float, min_lr: float, warmup_epochs: float,
                         epochs: float) -> float:
    """util/lr_sched.py:3-15 (float64 host math): linear warm-up, then half cosine."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (
        1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def absolute_lr(blr: float, eff_batch_size: int) -> float:
    """main_linprobe.py:572-573."""
    return blr * eff_batch_size / 256


def lars_step(params: List[np.ndarray], grads: List[Optional[np.ndarray]],
              mus: List[Optional[np.ndarray]], lr: float, weight_decay: float = 0.0,
              momentum: float = 0.9, trust_coefficient: float = 0.001):
    """util/lars.py:13-37.  ``params`` are updated in place-style (new arrays returned).
    Tensors with ndim <= 1 skip weight decay and the trust ratio (lars.py:21)."""
    new_p, new_mu = [], []
    for p, g, mu in zip(params, grads, mus):
        if g is None:
            new_p.append(p); new_mu.append(mu); continue
        dp = g.astype(F32)
        if p.ndim > 1:
            dp = (dp + F32(weight_decay) * p).astype(F32)
            pn = np.sqrt((p.astype(F32) ** 2).sum(dtype=F32))
            un = np.sqrt((dp ** 2).sum(dtype=F32))
            q = F32(trust_coefficient) * pn / un if (pn > 0 and un > 0) else F32(1.0)
            dp = (dp * F32(q)).astype(F32)
        if mu is None:
            mu = np.zeros_like(p, dtype=F32)
        mu = (mu * F32(momentum) + dp).astype(F32)
        new_mu.append(mu)
        new_p.append((p - F32(lr) * mu).astype(F32))
    return new_p, new_mu


def sgd_step(params, grads, lr: float, weight_decay: float = 0.0):
    """torch.optim.SGD(lr, weight_decay) with no momentum (main_linprobe.py:403-408)."""
    out = []
    for p, g in zip(params, grads):
        if g is None:
            out.append(p); continue
        d = (g + F32(weight_decay) * p).astype(F32) if weight_decay != 0 else g
        out.append((p - F32(lr) * d).astype(F32))
    return out


def adamw_step(params, grads, exp_avgs, exp_avg_sqs, step: int, lr: float,
               weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.AdamW defaults as selected by main_linprobe.py:403-408 (``step`` is the
    1-based step count AFTER this update)."""
    b1, b2 = betas
    out_p, out_m, out_v = [], [], []
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        p = (p * F32(1 - lr * weight_decay)).astype(F32)
        m = (m * F32(b1) + F32(1 - b1) * g).astype(F32)
        v = (v * F32(b2) + F32(1 - b2) * g * g).astype(F32)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)).astype(F32)
        p = (p - F32(lr / bc1) * (m / denom)).astype(F32)
        out_p.append(p); out_m.append(m); out_v.append(v)
    return out_p, out_m, out_v


def grad_norm(grads: Sequence[np.ndarray]) -> float:
    """util/misc.py:289-301 ``get_grad_norm_`` (L2 of per-tensor L2 norms)."""
    gs = [g for g in grads if g is not None]
    if not gs:
        return 0.0
    return float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in gs)))


@dataclass
class GradScalerState:
    """torch.cuda.amp.GradScaler defaults (util/misc.py:263-264): the semantics the native
    step reproduces -- scale the loss, unscale grads, skip the step on inf/nan, then
    grow/back off the scale."""
    scale: float = 65536.0
    growth_factor: float = 2.0
    backoff_factor: float = 0.5
    growth_interval: int = 2000
    growth_tracker: int = 0

    def update(self, found_inf: bool) -> None:
        if found_inf:
            self.scale *= self.backoff_factor
            self.growth_tracker = 0
        else:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0


# --------------------------------------------------------------------------- #
# The whole head: Sequential(EP, BN1d, Linear) + CE + optimizer
# (probe_heads.py:87-110, engine_finetune.py:52-77)
# --------------------------------------------------------------------------- #
PARAM_ORDER = ("cls_token", "v_weight", "fc_weight", "fc_bias")   # nn.Module.parameters() order


@dataclass
class HeadState:
    cls_token: np.ndarray      # (1,Q,C)          state_dict key 0.cls_token
    v_weight: np.ndarray       # (C//d_out, C)    0.v.weight
    fc_weight: np.ndarray      # (classes, C')    2.weight
    fc_bias: np.ndarray        # (classes,)       2.bias
    running_mean: np.ndarray   # (C',)            1.running_mean
    running_var: np.ndarray    # (C',)            1.running_var
    num_batches_tracked: int = 0
    num_queries: int = 1
    d_out: int = 1
    mu: Dict[str, Optional[np.ndarray]] = field(default_factory=dict)   # LARS state['mu']

    def params(self):
        return [getattr(self, k) for k in PARAM_ORDER]

    def set_params(self, ps):
        for k, p in zip(PARAM_ORDER, ps):
            setattr(self, k, p)


def head_forward_train(st: HeadState, x, targets):
    pooled, c_ep = ep_forward(x, st.cls_token, st.v_weight, st.num_queries, st.d_out)
    z, rm, rv, nbt, c_bn = bn_forward_train(pooled, st.running_mean, st.running_var,
                                            st.num_batches_tracked)
    logits = linear_forward(z, st.fc_weight, st.fc_bias)
    loss, dlogits = cross_entropy(logits, targets)
    cache = dict(ep=c_ep, bn=c_bn, z=z, dlogits=dlogits, new_bn=(rm, rv, nbt))
    return dict(pooled=pooled, z=z, logits=logits, loss=loss), cache


def head_backward(st: HeadState, cache, loss_scale: float = 1.0):
    dlogits = (cache["dlogits"] * F32(loss_scale)).astype(F32)
    dz, dW, db = linear_backward(dlogits, cache["z"], st.fc_weight)
    dy = bn_backward_train(dz, cache["bn"])
    dcls, dWv = ep_backward(dy, cache["ep"], st.v_weight)
    return dict(cls_token=dcls, v_weight=dWv, fc_weight=dW, fc_bias=db, dy=dy, dz=dz)


def head_forward_eval(st: HeadState, x):
    pooled, _ = ep_forward(x, st.cls_token, st.v_weight, st.num_queries, st.d_out)
    z = bn_forward_eval(pooled, st.running_mean, st.running_var)
    return linear_forward(z, st.fc_weight, st.fc_bias)


def head_forward_eval_fp16_autocast(st: HeadState, x):
    """The reference's evaluation forward as it actually runs: ``with torch.cuda.amp.autocast():`` wraps the model call
    in evaluate() (engine_finetune.py:131), i.e. fp16 autocast.  Autocast's op lists: Linear / matmul take fp16
    operands and return fp16 (fp32 accumulation inside), softmax runs in fp32 (and returns fp32), batch_norm takes the
    fp16 input, computes in fp32 and returns fp16.  Restated on the reference's own association (project every token,
    then pool -- ep.py:35-45).  Pinned on ``eval_logits_fp16_autocast`` of the golden fixtures."""
    r16 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)
    x = np.asarray(x, dtype=F32)
    B, N, C = x.shape
    Q, dq = st.num_queries, (C // st.d_out) // st.num_queries
    q = (st.cls_token[0] * F32(ep_scale(C))).astype(F32)                    # ep.py:37,39 (fp32 mul: not an autocast op)
    x16, w16 = r16(x), r16(st.v_weight)
    v = r16(x16.reshape(B * N, C) @ w16.T).reshape(B, N, Q, dq).transpose(0, 2, 1, 3)        # ep.py:40 self.v -> fp16
    s = r16(np.matmul(r16(q)[None], x16.transpose(0, 2, 1)))                # ep.py:42 q @ k^T -> fp16
    a = _softmax_lastdim(s)                                                 # ep.py:43 softmax: fp32
    pooled = r16(np.matmul(r16(a)[:, :, None, :], v)).reshape(B, C // st.d_out)                # ep.py:44 attn @ v -> fp16
    z = r16(bn_forward_eval(pooled, st.running_mean, st.running_var))       # probe_heads.py:110: fp32 math, fp16 out
    return r16(linear_forward(z, r16(st.fc_weight), r16(st.fc_bias)))       # probe_heads.py:76 Linear -> fp16


def head_train_step(st: HeadState, x, targets, lr: float, weight_decay: float = 0.0,
                    optimizer: str = "lars", momentum: float = 0.9,
                    trust_coefficient: float = 0.001):
    """One iteration of engine_finetune.py:40-77 for the head with fp32 (``--amp none``):
    forward, CE, backward, optimizer step, BN running-stat update."""
    out, cache = head_forward_train(st, x, targets)
    g = head_backward(st, cache)
    grads = [g[k] for k in PARAM_ORDER]
    if optimizer == "lars":
        mus = [st.mu.get(k) for k in PARAM_ORDER]
        ps, mus = lars_step(st.params(), grads, mus, lr, weight_decay, momentum, trust_coefficient)
        st.mu = dict(zip(PARAM_ORDER, mus))
    elif optimizer == "sgd":
        ps = sgd_step(st.params(), grads, lr, weight_decay)
    else:
        raise ValueError(optimizer)
    st.set_params(ps)
    st.running_mean, st.running_var, st.num_batches_tracked = cache["new_bn"]
    out["grads"] = g
    return out
