"""CBAM pooling (``--cls_features cbam``), native on MI355X.

Same constructor, parameter / buffer names and initialisation order as the reference ``CbamPooling`` with its ``ChannelAttn`` and
``SpatialAttn`` / ``ConvNormAct`` (reference poolings/cbam.py:19-139, poolings/clip/conv_bn_act.py:15-75), so reference
checkpoints load with ``strict=True`` (keys ``channel.fc1.weight`` (rd,C,1,1), ``channel.fc2.weight`` (C,rd,1,1),
``spatial.conv.conv.weight`` (1,2,k,k), ``spatial.conv.bn.weight``, ``.bias``, ``.running_mean``, ``.running_var``,
``.num_batches_tracked``) and a head built under ``torch.manual_seed(s)`` has bit-identical initial weights.

forward(x: (B, N, C)) -> (B, C) with N a perfect square.  On a GPU the head is a sequence of streaming passes over the tokens
(csrc/ep_cbam.hip).  Supported configuration = what the registry builds (reference probe_heads.py:77): reduction 1/16, ReLU,
sigmoid gates, no MLP bias, output_size 1.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn

from .. import functional as F_


class ChannelAttn(nn.Module):
    """Parameter container with the reference's names (cbam.py:19-31)."""

    def __init__(self, channels: int, rd_channels: int):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, rd_channels, 1, bias=False)
        self.act = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(rd_channels, channels, 1, bias=False)
        self.gate = nn.Sigmoid()


class _ConvNormAct(nn.Module):
    """``ConvNormAct(2, 1, k, apply_act=False)``: Conv2d (symmetric padding, no bias) + BatchNorm2d (conv_bn_act.py:43-66)."""

    def __init__(self, kernel_size: int):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size, stride=1, padding=(kernel_size - 1) // 2, bias=False)
        self.bn = nn.BatchNorm2d(1)


class SpatialAttn(nn.Module):
    def __init__(self, kernel_size: int = 7):
        super().__init__()
        self.conv = _ConvNormAct(kernel_size)
        self.gate = nn.Sigmoid()


class CbamPooling(nn.Module):
    def __init__(self, channels: int, rd_ratio: float = 1. / 16, rd_channels=None, rd_divisor: int = 1, spatial_kernel_size: int = 7,
                 act_layer=nn.ReLU, gate_layer="sigmoid", mlp_bias: bool = False, output_size: int = 1):
        super().__init__()
        if rd_channels or rd_divisor != 1 or act_layer is not nn.ReLU or gate_layer != "sigmoid" or mlp_bias or output_size != 1:
            raise NotImplementedError("native CBAM pooling supports the registry's configuration "
                                      "(CbamPooling(channels=dim, spatial_kernel_size=k))")
        if channels % 4 != 0 or spatial_kernel_size % 2 != 1:
            raise ValueError("channels must be a multiple of 4 and the spatial kernel size odd")
        rd = max(1, int(channels * rd_ratio + 0.5))                         # make_divisible(v, 1, round_limit=0.) (helpers.py)
        self.channel = ChannelAttn(channels, rd)                            # cbam.py:109-111 (fc1, fc2 draw first)
        self.spatial = SpatialAttn(spatial_kernel_size)                     # :112
        self.relu = nn.ReLU(inplace=True)
        self.avgpool = nn.AdaptiveAvgPool2d(output_size)
        self.rd, self.ks = rd, spatial_kernel_size

    def _tensors(self):
        return (self.channel.fc1.weight, self.channel.fc2.weight, self.spatial.conv.conv.weight, self.spatial.conv.bn.weight,
                self.spatial.conv.bn.bias)

    def forward(self, x: torch.Tensor, cls: Any = None, **_: Any) -> torch.Tensor:
        C_ = self.channel.fc1.in_channels
        if x.dim() != 3 or x.shape[-1] != C_:
            raise ValueError(f"expected tokens (B, N, {C_}), got {tuple(x.shape)}")
        side = int(x.shape[1] ** 0.5)
        if side * side != x.shape[1]:
            raise ValueError("n must be a perfect square for reshaping.")            # the reference's assertion (cbam.py:122)
        out_dtype = x.dtype
        bn = self.spatial.conv.bn
        if self.training and bn.momentum is None:
            raise NotImplementedError("native CBAM pooling: cumulative-average BatchNorm (momentum=None) is not supported")
        y = F_.cbam_pool(x, self.rd, self.ks, self.training, bn.eps, bn.momentum if bn.momentum is not None else 0.0,
                         bn.running_mean, bn.running_var, bn.num_batches_tracked if self.training else None, *self._tensors())
        return y if out_dtype == torch.float32 else y.to(out_dtype)
