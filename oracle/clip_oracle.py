"""CPU oracle for the CLIP attention-pooling probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features clip``: ``AttentionPool2d(in_features,
feat_size)`` (reference poolings/clip/attention_pool2d.py:100-169) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the
encoder's ``Linear`` (reference probe_heads.py:54-57,71,105-106).  It keeps the reference's association -- LayerNorm, prepend
the mean row, add the position embedding, one qkv Linear over all N + 1 rows, full (N + 1) x (N + 1) attention per head, proj,
row 0 -- and gradients come from autograd; it does NOT use the per-image derived-query algebra of the HIP path.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/clip_*.npz; tests/test_clip_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
from torch import nn


class ClipPort(nn.Module):
    def __init__(self, dim, n_tokens, num_heads=4):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5                               # attention_pool2d.py:131
        self.qkv = nn.Linear(dim, dim * 3)                               # :127
        self.proj = nn.Linear(dim, dim)                                  # :128
        self.pos_embed = nn.Parameter(torch.zeros(n_tokens + 1, dim))    # :133
        self.norm = nn.LayerNorm(dim, eps=1e-6)                          # :138

    def attention(self, x):
        B, N, d = x.shape
        x = self.norm(x)                                                 # :145
        x = torch.cat([x.mean(1, keepdim=True), x], dim=1)               # :150
        x = x + self.pos_embed.unsqueeze(0)                              # :151
        x = self.qkv(x).reshape(B, N + 1, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)   # :153
        q, k, v = x[0], x[1], x[2]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)                              # :155-156
        return attn, v

    def forward(self, x, cls=None):
        B, N, d = x.shape
        attn, v = self.attention(x)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N + 1, -1))[:, 0]                     # :158-159,169


def make_head(dim, nb_classes, n_tokens=196, num_heads=4):
    return nn.Sequential(ClipPort(dim, n_tokens, num_heads), nn.BatchNorm1d(dim, affine=False, eps=1e-6),
                         nn.Linear(dim, nb_classes))


PARAM_NAMES = ["pos_embed", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm_w", "norm_b", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.pos_embed, p.qkv.weight, p.qkv.bias, p.proj.weight, p.proj.bias, p.norm.weight, p.norm.bias, head[2].weight,
            head[2].bias]
