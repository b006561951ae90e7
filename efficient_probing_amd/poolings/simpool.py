"""SimPool heads (``--cls_features simpool`` / ``esimpool``), native on MI355X.

Same constructors, parameter names and initialisation order as the reference ``SimPool`` / ``SimPool_nolinears``
(reference poolings/simpool.py:5-91 / :93-170), so reference checkpoints load with ``strict=True`` (keys
``norm_patches.weight``, ``norm_patches.bias`` [, ``wq.weight``, ``wk.weight``]) and a head built under
``torch.manual_seed(s)`` has bit-identical initial weights.

forward(x: (B, N, d)) -> (B, d).  The query comes from the image's own mean token, keys are LayerNorm-ed (eps 1e-6) patch
tokens; on a GPU the heads run on the per-image-query token passes (csrc/ep_pool_imgq.hip, csrc/ep_simpool.hip).
Supported configuration = what the registry builds (reference probe_heads.py:66-70): no qkv bias, gamma=None; SimPool with
one head, SimPool_nolinears with any head count whose head width is a multiple of 32.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn

from .. import functional as F_


class _SimPoolBase(nn.Module):
    linears = True

    def _tensors(self):
        raise NotImplementedError

    def _check(self, x, cls):
        if cls is not None:
            raise NotImplementedError("native SimPool: query tokens from the caller (cls=) are not supported")
        if x.dim() == 4:
            raise NotImplementedError("native SimPool: CNN feature maps (B, d, H, W) are not supported; pass tokens (B, N, d)")
        if x.dim() != 3 or x.shape[-1] != self.norm_patches.normalized_shape[0]:
            raise ValueError(f"expected tokens (B, N, {self.norm_patches.normalized_shape[0]}), got {tuple(x.shape)}")

    def forward(self, x: torch.Tensor, cls: Any = None, return_attn: bool = False, **_: Any):
        self._check(x, cls)
        out_dtype = x.dtype
        if return_attn:
            with torch.no_grad():
                y, A = F_.simpool_attention(x, self.num_heads, self.linears, *self._tensors())
            return (y if out_dtype == torch.float32 else y.to(out_dtype)), A.unsqueeze(2)      # (B, H, 1, N) like the reference
        y = F_.simpool_pool(x, self.num_heads, self.linears, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)


class SimPool(_SimPoolBase):
    linears = True

    def __init__(self, dim: int, num_heads: int = 1, qkv_bias: bool = False, qk_scale=None, gamma=None, use_beta: bool = False):
        super().__init__()
        if qkv_bias or qk_scale is not None or gamma is not None or use_beta or num_heads != 1:
            raise NotImplementedError("native SimPool supports the registry's configuration "
                                      "(SimPool(dim, num_heads=1, qkv_bias=False, qk_scale=None, gamma=None))")
        if dim % 4 != 0:
            raise ValueError(f"dim={dim} must be a multiple of 4")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                  # simpool.py:9-10
        self.norm_patches = nn.LayerNorm(dim, eps=1e-6)          # simpool.py:12
        self.wq = nn.Linear(dim, dim, bias=qkv_bias)             # simpool.py:14
        self.wk = nn.Linear(dim, dim, bias=qkv_bias)             # simpool.py:15
        self.gamma, self.use_beta = gamma, use_beta

    def _tensors(self):
        return (self.norm_patches.weight, self.norm_patches.bias, self.wq.weight, self.wk.weight)


class SimPool_nolinears(_SimPoolBase):
    linears = False

    def __init__(self, dim: int, num_heads: int = 1, qkv_bias: bool = False, qk_scale=None, gamma=None, use_beta: bool = False):
        super().__init__()
        if qk_scale is not None or gamma is not None or use_beta:
            raise NotImplementedError("native SimPool_nolinears supports the registry's configuration "
                                      "(SimPool_nolinears(dim, num_heads=12, qk_scale=None, gamma=None))")
        if dim % num_heads != 0 or (dim // num_heads) % 32 != 0:
            raise ValueError(f"dim={dim} must split into {num_heads} heads of a multiple of 32 channels")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                  # simpool.py:97-98
        self.norm_patches = nn.LayerNorm(dim, eps=1e-6)          # simpool.py:100
        self.gamma, self.use_beta = gamma, use_beta

    def _tensors(self):
        return (self.norm_patches.weight, self.norm_patches.bias)
