"""DINOv2-block pooling (``--cls_features dinovit``), native on MI355X.

Same constructor, parameter names and initialisation order as the reference ``DinoViTBlockPooling`` (reference
poolings/other_pool.py:299-318) around one DINOv2 ``Block`` (poolings/dinov2_layers/block.py:43-113 with attention.py:37-69 and
mlp.py:17-41), so reference checkpoints load with ``strict=True`` (keys ``dino_block.norm1.weight`` / ``.bias``,
``dino_block.attn.qkv.weight`` (3D, D), ``dino_block.attn.proj.weight`` / ``.bias``, ``dino_block.norm2.*``,
``dino_block.mlp.fc1.*`` (4D, D), ``dino_block.mlp.fc2.*`` (D, 4D)) and a head built under ``torch.manual_seed(s)`` has
bit-identical initial weights.

forward(x: (B, N, D)) -> (B, D): ``(x1 + mlp(norm2(x1))).mean(1)`` with ``x1 = x + attn(norm1(x))``.  On a GPU every contraction
of the block runs on the exact-fp32 matrix-core kernel (csrc/ep_dinovit.hip).  Supported configuration = what the registry builds
(reference probe_heads.py:80): 8 heads, MLP ratio 4, no qkv bias, no LayerScale, no dropout / drop path.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn

from .. import functional as F_


class Attention(nn.Module):
    """Parameter container of the block's self-attention (attention.py:37-54)."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, proj_bias: bool = True, attn_drop: float = 0.0,
                 proj_drop: float = 0.0):
        super().__init__()
        if qkv_bias or not proj_bias or attn_drop or proj_drop:
            raise NotImplementedError("native DINOv2 block: qkv_bias=False, proj_bias=True, no dropout")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim, bias=True)


class Mlp(nn.Module):
    """Parameter container of the block's MLP (mlp.py:17-31)."""

    def __init__(self, in_features: int, hidden_features: int):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class DinoBlock(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = False, proj_bias: bool = True,
                 ffn_bias: bool = True, drop: float = 0.0, attn_drop: float = 0.0, init_values: Any = None, drop_path: float = 0.0):
        super().__init__()
        if not ffn_bias or drop or init_values or drop_path:
            raise NotImplementedError("native DINOv2 block: ffn_bias=True, no dropout, LayerScale or drop path")
        self.norm1 = nn.LayerNorm(dim)                                              # block.py:63
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias, attn_drop=attn_drop,
                              proj_drop=drop)                                       # :64-71
        self.norm2 = nn.LayerNorm(dim)                                              # :75
        self.mlp = Mlp(dim, int(dim * mlp_ratio))                                   # :76-83


class DinoViTBlockPooling(nn.Module):
    def __init__(self, d_model: int = 512, num_heads: int = 8):
        super().__init__()
        assert d_model % num_heads == 0, "d_model % num_heads should be zero."
        if (d_model // num_heads) % 4 != 0:
            raise ValueError(f"d_model / num_heads = {d_model // num_heads} must be a multiple of 4")
        self.dino_block = DinoBlock(dim=d_model, num_heads=num_heads)

    def _tensors(self):
        b = self.dino_block
        return (b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.proj.weight, b.attn.proj.bias, b.norm2.weight, b.norm2.bias,
                b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias)

    def forward(self, x: torch.Tensor, return_attention: bool = False, **_: Any):
        """``return_attention=True`` returns (pooled, attention (B, heads, N, N)); the reference's own forward fails there
        (other_pool.py:316-318 takes ``.mean`` of the (x, attention) tuple), so this is the intended result, not a mirrored one."""
        b = self.dino_block
        D = b.norm1.normalized_shape[0]
        if x.dim() != 3 or x.shape[-1] != D:
            raise ValueError(f"expected tokens (B, N, {D}), got {tuple(x.shape)}")
        if b.norm1.eps != b.norm2.eps:
            raise NotImplementedError("native DINOv2 block: both LayerNorms share one eps")
        out_dtype = x.dtype
        if return_attention:
            with torch.no_grad():
                y, A = F_.dinovit_attention(x, b.attn.num_heads, b.norm1.eps, *self._tensors())
            return (y if out_dtype == torch.float32 else y.to(out_dtype)), A
        y = F_.dinovit_pool(x, b.attn.num_heads, b.norm1.eps, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)

    def __repr__(self):
        return self.__class__.__name__
