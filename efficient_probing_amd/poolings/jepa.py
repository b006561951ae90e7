"""V-JEPA attentive pooler (``--cls_features jepa``), native on MI355X.

Same constructor, parameter names / shapes and initialisation (order of the random draws, truncated-normal recipe,
residual-branch rescaling) as the reference ``AttentivePooler`` with its ``CrossAttentionBlock`` / ``CrossAttention`` / ``MLP``
(reference poolings/jepa/attentive_pooler.py:21-104, poolings/jepa/modules.py:13-183, poolings/jepa/tensors.py:17-49), so a
head built under ``torch.manual_seed(s)`` has bit-identical initial weights and reference checkpoints load with
``strict=True``.

Supported configuration = what the registry builds (reference probe_heads.py:81: ``AttentivePooler(embed_dim=dim,
num_heads=args.num_heads)``): one query token, depth 1, complete block, qkv bias, LayerNorm.  On a GPU the head runs on the
LayerNorm-of-tokens mode of the EP streaming kernels (csrc/ep_siglip.hip, JEPA section); no CPU path here.
"""
from __future__ import annotations

import math
from typing import Any

import torch
from torch import nn

from .. import functional as F_


def _trunc_normal_(tensor: torch.Tensor, mean: float = 0.0, std: float = 1.0, a: float = -2.0, b: float = 2.0):
    """tensors.py:17-49: inverse-CDF truncated normal (bounds in units of the target distribution)."""
    def norm_cdf(x):
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0
    with torch.no_grad():
        lo, hi = norm_cdf((a - mean) / std), norm_cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1)
        tensor.erfinv_()
        tensor.mul_(std * math.sqrt(2.0))
        tensor.add_(mean)
        tensor.clamp_(min=a, max=b)
    return tensor


class MLP(nn.Module):
    def __init__(self, in_features: int, hidden_features: int):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(0.0)


class CrossAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int = 12, qkv_bias: bool = False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, int(dim * 2), bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_sdpa = True


class CrossAttentionBlock(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.xattn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(in_features=dim, hidden_features=int(dim * mlp_ratio))


class AttentivePooler(nn.Module):
    def __init__(self, num_queries: int = 1, embed_dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, depth: int = 1,
                 norm_layer=nn.LayerNorm, init_std: float = 0.02, qkv_bias: bool = True, complete_block: bool = True):
        super().__init__()
        if num_queries != 1 or depth != 1 or not complete_block or not qkv_bias or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("native JEPA pooler supports the registry's configuration "
                                      "(AttentivePooler(embed_dim=dim, num_heads=args.num_heads), reference probe_heads.py:81)")
        if embed_dim % num_heads != 0 or (embed_dim // num_heads) % 4 != 0:
            raise ValueError(f"embed_dim={embed_dim} must split into {num_heads} heads of a multiple of 4")
        self.query_tokens = nn.Parameter(torch.zeros(1, num_queries, embed_dim))
        self.complete_block = complete_block
        self.cross_attention_block = CrossAttentionBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                                         qkv_bias=qkv_bias)
        self.blocks = None
        self.init_std = init_std
        _trunc_normal_(self.query_tokens, std=self.init_std)                 # attentive_pooler.py:65
        self.apply(self._init_weights)                                       # :66
        self._rescale_blocks()                                               # :67

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=self.init_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _rescale_blocks(self):
        self.cross_attention_block.xattn.proj.weight.data.div_(math.sqrt(2.0))
        self.cross_attention_block.mlp.fc2.weight.data.div_(math.sqrt(2.0))

    def _tensors(self):
        b = self.cross_attention_block
        return (self.query_tokens, b.norm1.weight, b.norm1.bias, b.xattn.q.weight, b.xattn.q.bias, b.xattn.kv.weight,
                b.xattn.kv.bias, b.xattn.proj.weight, b.xattn.proj.bias, b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight,
                b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias)

    def forward(self, x: torch.Tensor, cls: Any = None, **_: Any) -> torch.Tensor:
        if cls is not None:
            raise NotImplementedError("native JEPA pooler: per-batch queries (cls=...) are not supported")
        b = self.cross_attention_block
        if x.dim() != 3 or x.shape[-1] != b.xattn.q.in_features:
            raise ValueError(f"expected tokens (B, N, {b.xattn.q.in_features}), got {tuple(x.shape)}")
        out_dtype = x.dtype
        y = F_.jepa_pool(x, b.xattn.num_heads, b.mlp.fc1.out_features, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)
