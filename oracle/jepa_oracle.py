"""CPU oracle for the V-JEPA attentive-pooler probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features jepa``: ``AttentivePooler(embed_dim, num_heads)``
(reference poolings/jepa/attentive_pooler.py:21-104) with ``CrossAttentionBlock`` / ``CrossAttention`` / ``MLP`` (reference
poolings/jepa/modules.py:13-183) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's ``Linear`` (reference
probe_heads.py:81,105-106).  Reference association: LayerNorm of every token, ``kv`` Linear over every token, per-head
attention, proj, residual, LayerNorm, MLP; gradients from autograd.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/jepa_*.npz; tests/test_jepa_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class JepaPort(nn.Module):
    def __init__(self, dim, num_heads=16, mlp_ratio=4.0):
        super().__init__()
        self.num_heads = num_heads
        self.query_tokens = nn.Parameter(torch.zeros(1, 1, dim))            # attentive_pooler.py:36
        self.norm1 = nn.LayerNorm(dim)                                      # modules.py:173
        self.q = nn.Linear(dim, dim)                                        # :135
        self.kv = nn.Linear(dim, dim * 2)                                   # :136
        self.proj = nn.Linear(dim, dim)                                     # :137
        self.norm2 = nn.LayerNorm(dim)                                      # :175
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))                     # :25
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)                     # :27

    def forward(self, x, cls=None):
        B, N, C = x.shape
        H = self.num_heads
        q0 = self.query_tokens.repeat(B, 1, 1)                              # attentive_pooler.py:99
        xn = self.norm1(x)                                                  # modules.py:180
        q = self.q(q0).reshape(B, 1, H, C // H).permute(0, 2, 1, 3)         # :142
        kv = self.kv(xn).reshape(B, N, 2, H, C // H).permute(2, 0, 3, 1, 4) # :145
        k, v = kv[0], kv[1]
        attn = ((q @ k.transpose(-2, -1)) * (C // H) ** -0.5).softmax(dim=-1)   # :150 (SDPA) == :152-153
        y = self.proj((attn @ v).transpose(1, 2).reshape(B, 1, C))          # :156-157
        q1 = q0 + y                                                         # :181
        q2 = q1 + self.fc2(F.gelu(self.fc1(self.norm2(q1))))                # :182
        return q2.squeeze(1)                                                # attentive_pooler.py:104


def make_head(dim, nb_classes, num_heads=16):
    return nn.Sequential(JepaPort(dim, num_heads), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["query", "n1_w", "n1_b", "q_w", "q_b", "kv_w", "kv_b", "proj_w", "proj_b", "n2_w", "n2_b", "fc1_w", "fc1_b",
               "fc2_w", "fc2_b", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.query_tokens, p.norm1.weight, p.norm1.bias, p.q.weight, p.q.bias, p.kv.weight, p.kv.bias, p.proj.weight,
            p.proj.bias, p.norm2.weight, p.norm2.bias, p.fc1.weight, p.fc1.bias, p.fc2.weight, p.fc2.bias, head[2].weight,
            head[2].bias]
