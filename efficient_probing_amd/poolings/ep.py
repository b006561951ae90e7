"""EfficientProbing pooling module, native on MI355X.

Same constructor, parameter names/shapes/initialisation ORDER and forward contract as the
reference module (reference poolings/ep.py:7-47): ``v = Linear(dim, dim // d_out, bias)`` is
created first (kaiming-uniform draw), then ``cls_token = randn(1, Q, dim) * 0.02`` -- so a
head built under ``torch.manual_seed(s)`` has bit-identical initial weights and a reference
checkpoint (keys ``v.weight``, ``cls_token``) loads with ``strict=True``.

forward(x: (B, N, C), cls=None) -> (B, C // d_out).  On a GPU the computation runs in the HIP
kernels (pool-then-project); there is no CPU implementation here.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import nn

from .. import functional as F_


class EfficientProbing(nn.Module):
    def __init__(self, dim: int, num_heads: int = 1, qkv_bias: bool = False,
                 qk_scale: Optional[float] = None, num_queries: int = 32, d_out: int = 1):
        super().__init__()
        if num_heads != 1:
            # the reference forward squeezes the head axis (ep.py:44) and is only meaningful for
            # one head; the registry never passes another value (probe_heads.py:75)
            raise NotImplementedError("native EfficientProbing supports num_heads == 1 (the registry default)")
        if qkv_bias:
            raise NotImplementedError("native EfficientProbing: value bias is not supported "
                                      "(reference default qkv_bias=False; pool-then-project needs a linear map)")
        if dim % (d_out * num_queries) != 0:
            raise ValueError(f"dim={dim} must be divisible by d_out*num_queries={d_out * num_queries}")
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.d_out = d_out
        self.num_queries = num_queries
        # creation order fixes the RNG stream: value projection first, then the queries
        self.v = nn.Linear(dim, dim // d_out, bias=False)
        self.cls_token = nn.Parameter(torch.randn(1, num_queries, dim) * 0.02)

    def forward(self, x: torch.Tensor, cls: Optional[torch.Tensor] = None, **_: Any) -> torch.Tensor:
        if x.dim() != 3 or x.shape[-1] != self.v.in_features:
            raise ValueError(f"expected tokens (B, N, {self.v.in_features}), got {tuple(x.shape)}")
        out_dtype = x.dtype
        if cls is not None:
            # per-image queries override the learned ones (reference ep.py:32-33); their gradient is one (Q, D) row block
            # per image (csrc: ep_pool_backward_per_image), cls_token takes none
            if tuple(cls.shape) != (x.shape[0], self.num_queries, x.shape[-1]):
                raise ValueError(f"cls must be (B, {self.num_queries}, {x.shape[-1]}) = one query set per image, got {tuple(cls.shape)}")
            y = F_.ep_pool_project(x, cls, self.v.weight, self.scale, per_image=True)
        else:
            y = F_.ep_pool_project(x, self.cls_token, self.v.weight, self.scale, per_image=False)
        return y if out_dtype == torch.float32 else y.to(out_dtype)

    @torch.no_grad()
    def attention(self, x: torch.Tensor) -> torch.Tensor:
        """softmax((cls_token * scale) @ x^T) as (B, Q, N) -- what reference
        tools/ep_attention_maps.py:52-58 recomputes from a saved head."""
        _, S, ML = F_.pool_forward(x, self.cls_token, self.scale)
        return F_.attention_from_scores(S, ML)
