"""CPU oracle for the CBAM probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features cbam``: ``CbamPooling(channels, spatial_kernel_size=7)``
with ``ChannelAttn`` / ``SpatialAttn`` (reference poolings/cbam.py:19-139) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the
encoder's ``Linear`` (reference probe_heads.py:77,105-106).  It keeps the reference's association -- reshape to the token grid,
channel MLP on the average- and max-pooled vectors, gate, channel mean / max maps, 7x7 convolution, BatchNorm2d, gate,
residual, ReLU, average pool -- and gradients come from autograd (amax splits its gradient evenly over ties).

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/cbam_*.npz; tests/test_cbam_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
from torch import nn


class CbamPort(nn.Module):
    def __init__(self, dim, ks=7):
        super().__init__()
        rd = max(1, int(dim / 16 + 0.5))                                   # cbam.py:25-26 make_divisible(dim / 16, 1, 0.)
        self.fc1 = nn.Conv2d(dim, rd, 1, bias=False)                       # :27
        self.fc2 = nn.Conv2d(rd, dim, 1, bias=False)                       # :29
        self.conv = nn.Conv2d(2, 1, ks, padding=(ks - 1) // 2, bias=False)     # :62 ConvNormAct(2, 1, ks, apply_act=False)
        self.bn = nn.BatchNorm2d(1)

    def forward(self, x, cls=None):
        B, N, C = x.shape
        H = W = int(N ** 0.5)
        x = x.permute(0, 2, 1).reshape(B, C, H, W)                         # :124
        residual = x
        x_avg = self.fc2(torch.relu(self.fc1(x.mean((2, 3), keepdim=True))))   # :34
        x_max = self.fc2(torch.relu(self.fc1(x.amax((2, 3), keepdim=True))))   # :35
        x = x * torch.sigmoid(x_avg + x_max)                               # :36
        a = torch.cat([x.mean(dim=1, keepdim=True), x.amax(dim=1, keepdim=True)], dim=1)   # :66
        x = x * torch.sigmoid(self.bn(self.conv(a)))                       # :67-68
        return torch.relu(x + residual).mean((2, 3))                       # :131-138


def make_head(dim, nb_classes):
    return nn.Sequential(CbamPort(dim), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["fc1_w", "fc2_w", "conv_w", "bn_w", "bn_b", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.fc1.weight, p.fc2.weight, p.conv.weight, p.bn.weight, p.bn.bias, head[2].weight, head[2].bias]
