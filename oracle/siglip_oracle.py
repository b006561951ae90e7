"""CPU oracle for the SigLIP attention-pool probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features siglip``: ``AttentionPoolLatent(dim)``
(reference poolings/clip/attention_pool.py:13-140; Mlp poolings/clip/mlp.py:13-49) behind ``BatchNorm1d(affine=False,
eps=1e-6)`` and the encoder's ``Linear`` (reference probe_heads.py:72,105-106).  It keeps the reference's association --
``kv(x)`` over every token with bias, per-head ``q k^T`` softmax, ``attn @ v``, proj, residual MLP -- and gradients come
from autograd on that graph; it does NOT use the derived-query / pool-then-project algebra of the HIP path
(csrc/ep_siglip.hip).

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/siglip_*.npz; tests/test_siglip_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file.
"""
from __future__ import annotations

import torch
from torch import nn


class SiglipPort(nn.Module):
    def __init__(self, dim, num_heads=8, mlp_ratio=4.0):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5                                   # attention_pool.py:40
        self.latent = nn.Parameter(torch.zeros(1, 1, dim))                   # :53
        self.q = nn.Linear(dim, dim)                                         # :55
        self.kv = nn.Linear(dim, dim * 2)                                    # :56
        self.proj = nn.Linear(dim, dim)                                      # :59
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))                      # mlp.py:34
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)                      # mlp.py:38

    def attention(self, x):
        B, N, C = x.shape
        q = self.q(self.latent.expand(B, -1, -1)).reshape(B, 1, self.num_heads, self.head_dim).transpose(1, 2)   # :106-107
        kv = self.kv(x).reshape(B, N, 2, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)                   # :109
        k, v = kv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)                                          # :120-122
        return attn, v

    def forward(self, x, cls=None):
        B, N, C = x.shape
        attn, v = self.attention(x)
        o = (attn @ v).transpose(1, 2).reshape(B, 1, C)                      # :123-124
        o = self.proj(o)                                                     # :125
        o = o + self.fc2(torch.nn.functional.gelu(self.fc1(o)))              # :128 (norm = Identity)
        return o[:, 0]                                                       # :131-132


def make_head(dim, nb_classes, num_heads=8):
    return nn.Sequential(SiglipPort(dim, num_heads), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["latent", "q_w", "q_b", "kv_w", "kv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "fc_weight",
               "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.latent, p.q.weight, p.q.bias, p.kv.weight, p.kv.bias, p.proj.weight, p.proj.bias, p.fc1.weight, p.fc1.bias,
            p.fc2.weight, p.fc2.bias, head[2].weight, head[2].bias]
