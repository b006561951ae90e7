"""CPU oracle for the SimPool probe heads -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features simpool`` (``SimPool(dim, num_heads=1)``,
reference poolings/simpool.py:5-91) and ``--cls_features esimpool`` (``SimPool_nolinears(dim, num_heads=12)``, :93-170)
behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's ``Linear`` (reference probe_heads.py:66-70,105-106).
It keeps the reference's association -- mean token, LayerNorm(eps 1e-6) of every patch token, wq / wk Linear (simpool),
per-head softmax, ``attn @ v`` -- and gradients come from autograd; it does NOT use the derived per-image query rows of
the HIP path.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/simpool_*.npz / esimpool_*.npz; tests/test_simpool_cpu.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import torch
from torch import nn


class SimPoolPort(nn.Module):
    def __init__(self, dim, num_heads=1, linears=True):
        super().__init__()
        self.num_heads, self.linears = num_heads, linears
        self.scale = (dim // num_heads) ** -0.5                     # simpool.py:9-10 / :97-98
        self.norm_patches = nn.LayerNorm(dim, eps=1e-6)            # :12 / :100
        if linears:
            self.wq = nn.Linear(dim, dim, bias=False)              # :14
            self.wk = nn.Linear(dim, dim, bias=False)              # :15

    def attention(self, x):
        B, N, d = x.shape
        H = self.num_heads
        gap = x.mean(-2).unsqueeze(1)                              # :30-31 / :118-119
        if self.linears:
            q, k, v = gap, self.norm_patches(x), self.norm_patches(x)                # :55
            q, k = self.wq(q), self.wk(k)                                            # :67-68
        else:
            q, k, v = self.norm_patches(gap), self.norm_patches(x), x                # :137
        qq = q.reshape(B, 1, H, d // H).permute(0, 2, 1, 3)
        kk = k.reshape(B, N, H, d // H).permute(0, 2, 1, 3)
        vv = v.reshape(B, N, H, d // H).permute(0, 2, 1, 3)
        attn = ((qq @ kk.transpose(-2, -1)) * self.scale).softmax(dim=-1)            # :73-75 / :154-156
        return attn, vv

    def forward(self, x, cls=None):
        B, N, d = x.shape
        attn, vv = self.attention(x)
        return (attn @ vv).transpose(1, 2).reshape(B, d)                             # :86 / :167 (+ squeeze)


def make_head(dim, nb_classes, linears=True, num_heads=None):
    H = (1 if linears else 12) if num_heads is None else num_heads
    return nn.Sequential(SimPoolPort(dim, H, linears), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


def param_names(linears):
    return ["norm_w", "norm_b"] + (["wq", "wk"] if linears else []) + ["fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.norm_patches.weight, p.norm_patches.bias] + ([p.wq.weight, p.wk.weight] if p.linears else []) + \
        [head[2].weight, head[2].bias]
