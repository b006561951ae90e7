"""SigLIP attention-pool head (``--cls_features siglip``), native on MI355X.

Same constructor, parameter names / shapes and initialisation ORDER as the reference ``AttentionPoolLatent``
(reference poolings/clip/attention_pool.py:13-140) with its ``Mlp`` (reference poolings/clip/mlp.py:13-49), so a head
built under ``torch.manual_seed(s)`` has bit-identical initial weights (the latent is drawn last, by the truncated-normal
recipe of reference poolings/clip/weight_init.py:8-38,70-96) and reference checkpoints load with ``strict=True``.

Supported configuration = what the registry builds (reference probe_heads.py:72: ``AttentionPoolLatent(in_features=dim)``):
one latent query, qkv bias, no q/k norm, no positional embedding, no output norm, pool ``'token'``; ``num_heads`` and
``mlp_ratio`` are free.  On a GPU the head runs on the EP streaming kernels (csrc/ep_siglip.hip); no CPU path here.
"""
from __future__ import annotations

import math
from typing import Any, Optional

import torch
from torch import nn

from .. import functional as F_


def _trunc_normal_tf_(tensor: torch.Tensor, mean: float = 0.0, std: float = 1.0, a: float = -2.0, b: float = 2.0):
    """weight_init.py:70-96: standard normal truncated to [a, b] by the inverse-CDF method, then scaled and shifted."""
    def norm_cdf(x):
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0
    with torch.no_grad():
        lo, hi = norm_cdf(a), norm_cdf(b)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1)
        tensor.erfinv_()
        tensor.mul_(math.sqrt(2.0))
        tensor.add_(0.0)
        tensor.clamp_(min=a, max=b)
        tensor.mul_(std).add_(mean)
    return tensor


class Mlp(nn.Module):
    """Parameter container with the reference's names (mlp.py:13-49): fc1, GELU, fc2."""

    def __init__(self, in_features: int, hidden_features: int):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop2 = nn.Dropout(0.0)


class AttentionPoolLatent(nn.Module):
    def __init__(self, in_features: int, out_features: Optional[int] = None, embed_dim: Optional[int] = None,
                 num_heads: int = 8, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 latent_len: int = 1, latent_dim: Optional[int] = None, pos_embed: str = "", pool_type: str = "token",
                 norm_layer=None, drop: float = 0.0):
        super().__init__()
        embed_dim = embed_dim or in_features
        out_features = out_features or in_features
        if (embed_dim != in_features or out_features != in_features or not qkv_bias or qk_norm or latent_len != 1
                or pos_embed or pool_type != "token" or norm_layer is not None or drop != 0.0
                or (latent_dim is not None and latent_dim != embed_dim)):
            raise NotImplementedError("native SigLIP attention pool supports the registry's configuration "
                                      "(AttentionPoolLatent(in_features=dim), reference probe_heads.py:72)")
        if embed_dim % num_heads != 0 or (embed_dim // num_heads) % 4 != 0:
            raise ValueError(f"embed_dim={embed_dim} must split into {num_heads} heads of a multiple of 4")
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.pool = pool_type
        self.pos_embed = None
        self.latent_dim = embed_dim
        self.latent_len = latent_len
        self.latent = nn.Parameter(torch.zeros(1, latent_len, embed_dim))       # attention_pool.py:53
        self.q = nn.Linear(embed_dim, embed_dim, bias=qkv_bias)
        self.kv = nn.Linear(embed_dim, embed_dim * 2, bias=qkv_bias)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.mlp = Mlp(embed_dim, int(embed_dim * mlp_ratio))
        self.init_weights()

    def init_weights(self):
        _trunc_normal_tf_(self.latent, std=self.latent_dim ** -0.5)              # attention_pool.py:68

    def _tensors(self):
        return (self.latent, self.q.weight, self.q.bias, self.kv.weight, self.kv.bias, self.proj.weight, self.proj.bias,
                self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias)

    def forward(self, x: torch.Tensor, return_attn: bool = False, cls: Any = None, **_: Any):
        if x.dim() != 3 or x.shape[-1] != self.q.in_features:
            raise ValueError(f"expected tokens (B, N, {self.q.in_features}), got {tuple(x.shape)}")
        out_dtype = x.dtype
        y = F_.siglip_pool(x, self.num_heads, self.mlp.fc1.out_features, *self._tensors())
        y = y if out_dtype == torch.float32 else y.to(out_dtype)
        if return_attn:
            return y, self.attention(x).unsqueeze(2)                             # (B, heads, 1, N)
        return y

    @torch.no_grad()
    def attention(self, x: torch.Tensor) -> torch.Tensor:
        return F_.siglip_attention(x, self.num_heads, self.mlp.fc1.out_features, *self._tensors())
