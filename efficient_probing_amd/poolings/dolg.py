"""DOLG spatial attention pooling (``--cls_features dolg``), native on MI355X.

Same constructor, parameter / buffer names and initialisation order as the reference ``SpatialAttention2d`` (reference
poolings/dolg/dolg.py:11-62 with the ResNet-style ``init_weights`` of poolings/dolg/net.py:16-21), so reference checkpoints load
with ``strict=True`` (keys ``conv1.weight`` (D,D,1,1), ``conv1.bias``, ``bn.weight``, ``bn.bias``, ``bn.running_mean``,
``bn.running_var``, ``bn.num_batches_tracked``, ``conv2.weight`` (1,D,1,1), ``conv2.bias``) and a head built under
``torch.manual_seed(s)`` has bit-identical initial weights.

forward(x: (B, N, D)) -> (B, D) with N a perfect square (the reference reshapes the tokens to an h x w grid).  On a GPU the
head runs on the exact-fp32 matrix-core contraction, the BatchNorm kernels over all token rows and one streaming row kernel per
direction (csrc/ep_dolg.hip).  Supported configuration = what the registry builds (reference probe_heads.py:82): no ASPP,
s3_dim = in_c, ReLU.
"""
from __future__ import annotations

import math
from typing import Any

import torch
from torch import nn

from .. import functional as F_


def _init_conv(m: nn.Conv2d) -> None:
    fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels              # net.py:20-21
    m.weight.data.normal_(mean=0.0, std=math.sqrt(2.0 / fan_out))


class SpatialAttention2d(nn.Module):
    def __init__(self, in_c: int, s3_dim: int = 1024, act_fn: str = "relu", with_aspp: bool = False, bn_eps: float = 1e-5,
                 bn_nom: float = 0.1):
        super().__init__()
        if with_aspp or s3_dim != in_c or act_fn.lower() != "relu":
            raise NotImplementedError("native DOLG pooling supports the registry's configuration "
                                      "(SpatialAttention2d(in_c=dim, s3_dim=dim, with_aspp=False))")
        if in_c % 4 != 0:
            raise ValueError(f"in_c={in_c} must be a multiple of 4")
        self.with_aspp = with_aspp
        self.conv1 = nn.Conv2d(in_c, s3_dim, 1, 1)                               # dolg.py:21
        self.bn = nn.BatchNorm2d(s3_dim, eps=bn_eps, momentum=bn_nom)            # :22
        self.act1 = nn.ReLU()
        self.conv2 = nn.Conv2d(s3_dim, 1, 1, 1)                                  # :27
        self.softplus = nn.Softplus(beta=1, threshold=20)
        for conv in (self.conv1, self.conv2):                                    # :30-31
            _init_conv(conv)

    def _tensors(self):
        return (self.conv1.weight, self.conv1.bias, self.bn.weight, self.bn.bias, self.conv2.weight, self.conv2.bias)

    def _check(self, x, cls):
        if cls is not None:
            raise NotImplementedError("native DOLG pooling takes the tokens only")
        D = self.conv1.in_channels
        if x.dim() != 3 or x.shape[-1] != D:
            raise ValueError(f"expected tokens (B, N, {D}), got {tuple(x.shape)}")
        side = int(x.shape[1] ** 0.5)
        if side * side != x.shape[1]:
            raise ValueError(f"N = {x.shape[1]} is not a square token grid (the reference's view(b, c, h, w) fails too)")

    def forward(self, x: torch.Tensor, cls: Any = None, block_attmaps: Any = None, return_attn: bool = False, **_: Any):
        self._check(x, cls)
        out_dtype = x.dtype
        bn = self.bn
        if return_attn:
            with torch.no_grad():
                y, att = F_.dolg_attention(x, self.training, bn.eps, bn.running_mean, bn.running_var, *self._tensors())
            side = int(x.shape[1] ** 0.5)
            return (y if out_dtype == torch.float32 else y.to(out_dtype)), att.view(x.shape[0], 1, side, side)
        if self.training and bn.momentum is None:
            raise NotImplementedError("native DOLG pooling: cumulative-average BatchNorm (momentum=None) is not supported")
        y = F_.dolg_pool(x, self.training, bn.eps, bn.momentum if bn.momentum is not None else 0.0, bn.running_mean,
                         bn.running_var, bn.num_batches_tracked if self.training else None, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)

    def __repr__(self):
        return self.__class__.__name__
