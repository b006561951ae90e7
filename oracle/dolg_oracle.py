"""CPU oracle for the DOLG spatial-attention probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features dolg``: ``SpatialAttention2d(in_c, s3_dim=in_c,
with_aspp=False)`` (reference poolings/dolg/dolg.py:11-62) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's
``Linear`` (reference probe_heads.py:82,105-106).  It keeps the reference's association -- reshape to the token grid, 1x1
Conv2d, BatchNorm2d, L2 normalisation over channels, ReLU, 1x1 Conv2d to one channel, Softplus, product, mean over positions
-- and gradients come from autograd.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/dolg_*.npz; tests/test_dolg_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


class DolgPort(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, 1, 1)                     # dolg.py:21
        self.bn = nn.BatchNorm2d(dim, eps=1e-5, momentum=0.1)      # :22
        self.conv2 = nn.Conv2d(dim, 1, 1, 1)                       # :27

    def scores(self, x):
        b, hw, c = x.shape
        h = w = int(hw ** 0.5)
        x = x.permute(0, 2, 1).contiguous().view(b, c, h, w)       # :39-42
        x = self.bn(self.conv1(x))                                 # :47-48
        fmn = F.normalize(x, p=2, dim=1)                           # :50
        att = F.softplus(self.conv2(F.relu(x)), beta=1, threshold=20)   # :52-55
        return att, fmn

    def forward(self, x, cls=None):
        b, hw, c = x.shape
        att, fmn = self.scores(x)
        return (att.expand_as(fmn) * fmn).view(b, c, -1).permute(0, 2, 1).mean(1)   # :56-57,60


def make_head(dim, nb_classes):
    return nn.Sequential(DolgPort(dim), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["conv1_w", "conv1_b", "bn_w", "bn_b", "conv2_w", "conv2_b", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.conv1.weight, p.conv1.bias, p.bn.weight, p.bn.bias, p.conv2.weight, p.conv2.bias, head[2].weight, head[2].bias]
