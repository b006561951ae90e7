"""CLIP attention pooling (``--cls_features clip``), native on MI355X.

Same constructor, parameter names / shapes and initialisation order as the reference ``AttentionPool2d`` (reference
poolings/clip/attention_pool2d.py:100-169), so reference checkpoints load with ``strict=True`` (keys ``pos_embed``,
``qkv.weight``, ``qkv.bias``, ``proj.weight``, ``proj.bias``, ``norm.weight``, ``norm.bias``) and a head built under
``torch.manual_seed(s)`` has bit-identical initial weights (qkv and proj draw their Linear init first, then the truncated
normals of pos_embed and qkv.weight).

forward(x: (B, N, d)) -> (B, d) with N == feat_size ** 2 (the learned position embedding fixes the token count).  On a GPU
the head runs on the token passes with per-image query rows, an additive position score bias and the mean row merged in as
an extra softmax entry (csrc/ep_clip.hip).  Supported configuration = what the registry builds (reference
probe_heads.py:54-57): embed_dim = out_features = in_features, qkv bias.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn

from .. import functional as F_


class AttentionPool2d(nn.Module):
    def __init__(self, in_features: int, feat_size, out_features: int = None, embed_dim: int = None, num_heads: int = 4,
                 qkv_bias: bool = True):
        super().__init__()
        if (out_features or in_features) != in_features or (embed_dim or in_features) != in_features or not qkv_bias:
            raise NotImplementedError("native CLIP pooling supports the registry's configuration "
                                      "(AttentionPool2d(in_features=dim, feat_size=s))")
        if in_features % num_heads != 0 or (in_features // num_heads) % 4 != 0:
            raise ValueError(f"in_features={in_features} must split into {num_heads} heads of a multiple of 4")
        self.feat_size = (feat_size, feat_size) if isinstance(feat_size, int) else tuple(feat_size)
        spatial_dim = self.feat_size[0] * self.feat_size[1]
        if spatial_dim % 4 != 0:
            raise ValueError(f"feat_size {self.feat_size}: the token count must be a multiple of 4")
        self.qkv = nn.Linear(in_features, in_features * 3, bias=qkv_bias)          # attention_pool2d.py:127
        self.proj = nn.Linear(in_features, in_features)                            # :128
        self.num_heads = num_heads
        self.head_dim = in_features // num_heads
        self.scale = self.head_dim ** -0.5
        self.pos_embed = nn.Parameter(torch.zeros(spatial_dim + 1, in_features))   # :133
        nn.init.trunc_normal_(self.pos_embed, std=in_features ** -0.5)             # :134
        nn.init.trunc_normal_(self.qkv.weight, std=in_features ** -0.5)            # :135
        nn.init.zeros_(self.qkv.bias)                                              # :136
        self.norm = nn.LayerNorm(in_features, eps=1e-6)                            # :138

    def _tensors(self):
        return (self.pos_embed, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.norm.weight,
                self.norm.bias)

    def forward(self, x: torch.Tensor, cls: Any = None, return_attn: bool = False, **_: Any):
        if cls is not None:
            raise NotImplementedError("native CLIP pooling: a query row from the caller (cls=) is not supported")
        D = self.norm.normalized_shape[0]
        if x.dim() != 3 or x.shape[-1] != D or x.shape[1] + 1 != self.pos_embed.shape[0]:
            raise ValueError(f"expected tokens (B, {self.pos_embed.shape[0] - 1}, {D}), got {tuple(x.shape)} "
                             f"(the position embedding fixes the token count, as in the reference)")
        out_dtype = x.dtype
        if return_attn:
            with torch.no_grad():
                y, A = F_.clip_attention(x, self.num_heads, *self._tensors())
            return (y if out_dtype == torch.float32 else y.to(out_dtype)), A
        y = F_.clip_pool(x, self.num_heads, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)
