"""CAE attentive block (``--cls_features cae``), native on MI355X.

Same constructor, parameter names / shapes and initialisation order as the reference ``CAEAttentiveBlock`` with its
``CrossAttention`` (reference poolings/cae_att.py:19-108), so reference checkpoints load with ``strict=True`` and a head
built under ``torch.manual_seed(s)`` has bit-identical initial weights.

forward(x_kv: (B, N, C)) -> (B, C).  The keys and values are two LayerNorms of the same token; on a GPU the head runs on
the LayerNorm-of-tokens mode of the EP streaming kernels (csrc/ep_coca.hip, CAE section).  Supported configuration = what
the registry builds (reference probe_heads.py:83: ``CAEAttentiveBlock(dim=dim)``): no qkv bias, no positional terms.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn

from .. import functional as F_


class CrossAttention(nn.Module):
    """Parameter container with the reference's names (cae_att.py:19-46)."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False):
        super().__init__()
        if qkv_bias:
            raise NotImplementedError("native CAE block: qkv_bias=False as built by the registry")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=False)
        self.k = nn.Linear(dim, dim, bias=False)
        self.v = nn.Linear(dim, dim, bias=False)
        self.q_bias = None
        self.k_bias = None
        self.v_bias = None
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)


class CAEAttentiveBlock(nn.Module):
    def __init__(self, dim: int, num_heads: int = 8, mlp_ratio: float = 4.0, qkv_bias: bool = False, qk_scale=None,
                 drop: float = 0.0, attn_drop: float = 0.0, drop_path: float = 0.0, init_values=None, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, window_size=None, attn_head_dim=None):
        super().__init__()
        if (qkv_bias or qk_scale is not None or drop or attn_drop or drop_path or attn_head_dim is not None
                or norm_layer is not nn.LayerNorm):
            raise NotImplementedError("native CAE block supports the registry's configuration (CAEAttentiveBlock(dim=dim))")
        if dim % num_heads != 0 or (dim // num_heads) % 4 != 0:
            raise ValueError(f"dim={dim} must split into {num_heads} heads of a multiple of 4")
        self.query_token = nn.Parameter(torch.zeros(1, 1, dim))          # cae_att.py:86 (stays zero at init)
        self.norm1_q = norm_layer(dim)
        self.norm1_k = norm_layer(dim)
        self.norm1_v = norm_layer(dim)
        self.norm2_cross = norm_layer(dim)                               # created by the reference, unused by its forward
        self.cross_attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = nn.Identity()

    def _tensors(self):
        c = self.cross_attn
        return (self.query_token, self.norm1_q.weight, self.norm1_q.bias, self.norm1_k.weight, self.norm1_k.bias,
                self.norm1_v.weight, self.norm1_v.bias, self.norm2_cross.weight, self.norm2_cross.bias, c.q.weight,
                c.k.weight, c.v.weight, c.proj.weight, c.proj.bias)

    def forward(self, x_kv: torch.Tensor, pos_q=0, pos_k=0, cls: Any = None, **_: Any) -> torch.Tensor:
        if not (isinstance(pos_q, (int, float)) and pos_q == 0 and isinstance(pos_k, (int, float)) and pos_k == 0):
            raise NotImplementedError("native CAE block: positional terms are not supported (the probe passes none)")
        if x_kv.dim() != 3 or x_kv.shape[-1] != self.cross_attn.q.in_features:
            raise ValueError(f"expected tokens (B, N, {self.cross_attn.q.in_features}), got {tuple(x_kv.shape)}")
        out_dtype = x_kv.dtype
        y = F_.cae_pool(x_kv, self.cross_attn.num_heads, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)
