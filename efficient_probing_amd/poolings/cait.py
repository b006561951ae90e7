"""CaiT class-attention pooling (``--cls_features cait``), native on MI355X.

Same constructor, parameter names / shapes and initialisation order as the reference ``CAPooling`` with its
``LayerScale_Block_CA`` / ``Class_Attention`` / ``Mlp`` (reference poolings/other_pool.py:390-507, poolings/clip/mlp.py:13-50),
so reference checkpoints load with ``strict=True`` and a head built under ``torch.manual_seed(s)`` has bit-identical initial
weights (the Linear layers draw first, in the order q, k, v, proj, fc1, fc2; the class token's truncated normal last).

forward(x: (B, N, D)) -> (B, D).  On a GPU the head runs on the LayerNorm-of-tokens mode of the EP token passes with the
class row merged in as one extra softmax entry (csrc/ep_cait.hip).  Supported configuration = what the registry builds
(reference probe_heads.py:79: ``CAPooling(embed_dim=dim)``): one class-attention block, no dropout / drop-path.
"""
from __future__ import annotations

from functools import partial
from typing import Any

import torch
from torch import nn

from .. import functional as F_


class Class_Attention(nn.Module):
    """Parameter container with the reference's names (other_pool.py:440-454)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)


class Mlp(nn.Module):
    """Parameter container with the reference's names (poolings/clip/mlp.py:13-40)."""

    def __init__(self, in_features, hidden_features, act_layer=nn.GELU):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop2 = nn.Dropout(0.0)


class LayerScale_Block_CA(nn.Module):
    """Parameter container with the reference's names and construction order (other_pool.py:477-495)."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer, init_values):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Class_Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
        self.gamma_2 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)


class CAPooling(nn.Module):
    def __init__(self, embed_dim=512, num_heads=4, iterations=1, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 act_layer=nn.GELU, qk_scale=None, init_scale=1e-5, mlp_ratio_clstk=4.0):
        super().__init__()
        if iterations != 1 or not qkv_bias or qk_scale is not None or act_layer is not nn.GELU:
            raise NotImplementedError("native CaiT pooling supports the registry's configuration (CAPooling(embed_dim=dim))")
        probe = norm_layer(embed_dim)
        if not isinstance(probe, nn.LayerNorm) or abs(probe.eps - 1e-6) > 1e-12:
            raise NotImplementedError("native CaiT pooling: norm_layer must be LayerNorm(eps=1e-6)")
        hidden = int(embed_dim * mlp_ratio_clstk)
        if embed_dim % num_heads != 0 or (embed_dim // num_heads) % 4 != 0 or hidden % 4 != 0:
            raise ValueError(f"embed_dim={embed_dim} must split into {num_heads} heads of a multiple of 4")
        self.depth_token_only = iterations
        self.blocks_token_only = nn.ModuleList([
            LayerScale_Block_CA(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio_clstk, qkv_bias=qkv_bias,
                                norm_layer=norm_layer, init_values=init_scale) for _ in range(iterations)])
        self.norm = nn.LayerNorm(embed_dim)                                    # other_pool.py:415 (eps 1e-5)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))            # :416
        nn.init.trunc_normal_(self.cls_token, std=.02)                         # :417 (timm's trunc_normal_: a=-2, b=2)
        self.num_heads, self.hidden = num_heads, hidden

    def _tensors(self):
        b = self.blocks_token_only[0]
        a, m = b.attn, b.mlp
        return (self.cls_token, b.gamma_1, b.gamma_2, b.norm1.weight, b.norm1.bias, a.q.weight, a.q.bias, a.k.weight, a.k.bias,
                a.v.weight, a.v.bias, a.proj.weight, a.proj.bias, b.norm2.weight, b.norm2.bias, m.fc1.weight, m.fc1.bias,
                m.fc2.weight, m.fc2.bias, self.norm.weight, self.norm.bias)

    def forward(self, x: torch.Tensor, cls: Any = None, **_: Any) -> torch.Tensor:
        if cls is not None:
            raise NotImplementedError("native CaiT pooling: class tokens from the caller (cls=) are not supported")
        D = self.norm.normalized_shape[0]
        if x.dim() != 3 or x.shape[-1] != D:
            raise ValueError(f"expected tokens (B, N, {D}), got {tuple(x.shape)}")
        out_dtype = x.dtype
        y = F_.cait_pool(x, self.num_heads, self.hidden, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)
