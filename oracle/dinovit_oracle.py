"""CPU oracle for the DINOv2-block probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features dinovit``: ``DinoViTBlockPooling(d_model=dim)``
(reference poolings/other_pool.py:299-318: one poolings/dinov2_layers/block.py:43-113 Block with 8 heads, then the mean over
the tokens) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's ``Linear`` (reference probe_heads.py:80,105-106).
It keeps the reference's association -- LayerNorm, one (3D x D) qkv projection without bias split into heads, q scaled by
head_dim^-0.5 BEFORE the q k^T product (attention.py:49), softmax, attn @ v, projection, residual; LayerNorm, fc1, exact GELU,
fc2, residual; mean(dim=1) -- and gradients come from autograd.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/dinovit_*.npz; tests/test_dinovit_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


class DinovitPort(nn.Module):
    def __init__(self, dim, num_heads=8, mlp_ratio=4.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                    # attention.py:48-49
        self.norm1 = nn.LayerNorm(dim)                             # block.py:63
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)             # attention.py:51
        self.proj = nn.Linear(dim, dim)                            # :53
        self.norm2 = nn.LayerNorm(dim)                             # block.py:75
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))            # mlp.py:29
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)            # :31

    def attention(self, h):
        B, N, C = h.shape
        qkv = self.qkv(h).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)   # attention.py:58
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]                                                     # :60
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                                                  # :61-63
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C)), attn                               # :66-67

    def forward(self, x, return_attention=False):
        a, attn = self.attention(self.norm1(x))
        x = x + a                                                  # block.py:109
        x = x + self.fc2(F.gelu(self.fc1(self.norm2(x))))          # :110, mlp.py:35-40
        out = x.mean(dim=1)                                        # other_pool.py:318
        return (out, attn) if return_attention else out


def make_head(dim, nb_classes):
    return nn.Sequential(DinovitPort(dim), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["n1_w", "n1_b", "qkv_w", "proj_w", "proj_b", "n2_w", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "fc_weight",
               "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.norm1.weight, p.norm1.bias, p.qkv.weight, p.proj.weight, p.proj.bias, p.norm2.weight, p.norm2.bias, p.fc1.weight,
            p.fc1.bias, p.fc2.weight, p.fc2.bias, head[2].weight, head[2].bias]
