"""CPU oracle for the AIM attention-pooling probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features aim``: ``AttentionPoolingClassifier(dim,
num_heads)`` (reference poolings/aim.py:337-392) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's
``Linear`` (reference probe_heads.py:73,105-106).  It keeps the reference's association -- BatchNorm1d over the
transposed tokens, k / v Linear over every normalised token, per-head softmax, ``attn @ v`` -- and gradients come from
autograd; it does NOT use the derived-query / folded-projection algebra of the HIP path.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/aim_*.npz; tests/test_aim_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
from torch import nn


class AimPort(nn.Module):
    def __init__(self, dim, num_heads=12):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                     # aim.py:349-350
        self.k = nn.Linear(dim, dim, bias=False)                   # :352
        self.v = nn.Linear(dim, dim, bias=False)                   # :353
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim) * 0.02)   # :355
        self.bn = nn.BatchNorm1d(dim, affine=False, eps=1e-6)      # :357

    def attention(self, x):
        B, N, C = x.shape
        H = self.num_heads
        x = self.bn(x.transpose(-2, -1)).transpose(-2, -1)         # :364
        q = self.cls_token.expand(B, -1, -1).reshape(B, 1, H, C // H).permute(0, 2, 1, 3) * self.scale   # :369-380
        k = self.k(x).reshape(B, N, H, C // H).permute(0, 2, 1, 3)                                       # :374-378
        v = self.v(x).reshape(B, N, H, C // H).permute(0, 2, 1, 3)                                       # :381-385
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                                                 # :387-388
        return attn, v

    def forward(self, x, cls=None):
        B, N, C = x.shape
        attn, v = self.attention(x)
        return (attn @ v).transpose(1, 2).reshape(B, 1, C).mean(dim=1)                                   # :390-392


def make_head(dim, nb_classes, num_heads=16):
    return nn.Sequential(AimPort(dim, num_heads), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["cls_token", "k_w", "v_w", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.cls_token, p.k.weight, p.v.weight, head[2].weight, head[2].bias]
