"""AIM attention-pooling head (``--cls_features aim``), native on MI355X.

Same constructor, parameter / buffer names and initialisation order as the reference ``AttentionPoolingClassifier``
(reference poolings/aim.py:337-392), so reference checkpoints load with ``strict=True`` (keys ``k.weight``, ``v.weight``,
``cls_token``, ``bn.running_mean``, ``bn.running_var``, ``bn.num_batches_tracked``) and a head built under
``torch.manual_seed(s)`` has bit-identical initial weights.

forward(x: (B, N, C)) -> (B, C).  Every token is batch-normalised per channel (train: statistics of the B*N tokens of
the batch, running statistics updated; eval: running statistics), keys / values are linear maps of the normalised
tokens and one learned query token (H heads) attends over them.  On a GPU this runs on the plain EP token passes
(csrc/ep_aim.hip).  Supported configuration = what the registry builds (reference probe_heads.py:73): no qkv bias, one
query.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import nn

from .. import functional as F_


class AttentionPoolingClassifier(nn.Module):
    def __init__(self, dim: int, num_heads: int = 12, qkv_bias: bool = False, qk_scale: Optional[float] = None,
                 num_queries: int = 1):
        super().__init__()
        if qkv_bias or qk_scale is not None or num_queries != 1:
            raise NotImplementedError("native AIM head supports the registry's configuration "
                                      "(AttentionPoolingClassifier(dim=dim, num_heads=H))")
        if dim % num_heads != 0 or (dim // num_heads) % 4 != 0:
            raise ValueError(f"dim={dim} must split into {num_heads} heads of a multiple of 4")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                                  # aim.py:349-350
        self.k = nn.Linear(dim, dim, bias=qkv_bias)                              # aim.py:352 (drawn first)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)                              # aim.py:353
        self.cls_token = nn.Parameter(torch.randn(1, num_queries, dim) * 0.02)   # aim.py:355
        self.bn = nn.BatchNorm1d(dim, affine=False, eps=1e-6)                    # aim.py:357
        self.num_queries = num_queries

    def _tensors(self):
        return (self.cls_token, self.k.weight, self.v.weight)

    def forward(self, x: torch.Tensor, cls: Any = None, **_: Any) -> torch.Tensor:
        if cls is not None:
            raise NotImplementedError("native AIM head: per-batch query tokens (cls=) are not supported")
        if x.dim() != 3 or x.shape[-1] != self.k.in_features:
            raise ValueError(f"expected tokens (B, N, {self.k.in_features}), got {tuple(x.shape)}")
        out_dtype = x.dtype
        bn = self.bn
        if self.training and bn.momentum is None:
            raise NotImplementedError("native AIM head: cumulative-average BatchNorm (momentum=None) is not supported")
        y = F_.aim_pool(x, self.num_heads, self.training, bn.eps, bn.momentum if bn.momentum is not None else 0.0,
                        bn.running_mean, bn.running_var, bn.num_batches_tracked if self.training else None,
                        *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)

    @torch.no_grad()
    def attention(self, x: torch.Tensor) -> torch.Tensor:
        """(B, H, N) attention weights of the query token (reference aim.py:387-388)."""
        return F_.aim_attention(x, self.num_heads, self.training, self.bn.eps, self.bn.running_mean, self.bn.running_var,
                                *self._tensors())
