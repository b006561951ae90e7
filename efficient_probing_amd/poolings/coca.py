"""CoCa attentional pooler (``--cls_features coca``), native on MI355X.

Same constructor, parameter / buffer names, shapes and initialisation ORDER as the reference module
(reference poolings/coca_pytorch.py:250-343 ``CrossAttention`` and its bias-free ``LayerNorm`` :70-77), so a head
built under ``torch.manual_seed(s)`` has bit-identical initial weights and reference checkpoints
(keys ``norm.gamma``, ``norm.beta``, ``img_queries``, ``to_q.weight``, ``to_kv.weight``, ``to_out.weight``)
load with ``strict=True``.

forward(context: (B, N, C), cls=None) -> (B, C): ``out[:, 0]`` of the reference, i.e. image query 0 attending
over the tokens with ``heads`` query heads and ONE shared key/value head.  On a GPU this runs on the EP
streaming kernels (see csrc/ep_coca.hip for the algebra); there is no CPU implementation here.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import nn

from .. import functional as F_


class LayerNorm(nn.Module):
    """LayerNorm with a learned gain and a constant zero bias kept as a buffer (coca_pytorch.py:70-77)."""

    def __init__(self, dim: int):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:      # only used off the native path (e.g. CPU inspection)
        return torch.nn.functional.layer_norm(x, x.shape[-1:], self.gamma, self.beta)


class CrossAttention(nn.Module):
    def __init__(self, dim: int, *, context_dim: Optional[int] = None, dim_head: int = 64, num_img_queries: int = 196,
                 heads: int = 8, parallel_ff: bool = False, ff_mult: int = 4, norm_context: bool = False):
        super().__init__()
        if parallel_ff or norm_context:
            raise NotImplementedError("native CoCa pooler: parallel_ff / norm_context are not supported "
                                      "(the probe registry builds CrossAttention(dim=dim), reference probe_heads.py:78)")
        if context_dim is not None and context_dim != dim:
            raise NotImplementedError("native CoCa pooler: context_dim must equal dim")
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        inner_dim = heads * dim_head
        # creation order fixes the RNG stream (coca_pytorch.py:271-278)
        self.norm = LayerNorm(dim)
        self.context_norm = nn.Identity()
        self.img_queries = nn.Parameter(torch.randn(num_img_queries, dim))
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self.ff = None

    def _tensors(self):
        return (self.norm.gamma, self.norm.beta, self.img_queries, self.to_q.weight, self.to_kv.weight,
                self.to_out.weight)

    def forward(self, context: torch.Tensor, cls: Optional[torch.Tensor] = None, **_: Any) -> torch.Tensor:
        if cls is not None:
            raise NotImplementedError("native CoCa pooler: per-batch queries (cls=...) are not supported; "
                                      "the probe path never passes them")
        if context.dim() != 3 or context.shape[-1] != self.to_q.in_features:
            raise ValueError(f"expected tokens (B, N, {self.to_q.in_features}), got {tuple(context.shape)}")
        out_dtype = context.dtype
        y = F_.coca_pool(context, *self._tensors(), self.heads, self.dim_head)
        return y if out_dtype == torch.float32 else y.to(out_dtype)

    @torch.no_grad()
    def attention(self, context: torch.Tensor) -> torch.Tensor:
        """softmax attention of image query 0, (B, heads, N)."""
        return F_.coca_attention(context, *self._tensors(), self.heads, self.dim_head)
