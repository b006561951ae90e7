"""CPU oracle for the CAE attentive-block probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features cae``: ``CAEAttentiveBlock(dim)`` with its
``CrossAttention`` (reference poolings/cae_att.py:19-108) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's
``Linear`` (reference probe_heads.py:83,105-106).  It keeps the reference's association -- three LayerNorms, q / k / v
Linear over every token, per-head softmax, ``attn @ v``, proj -- and gradients come from autograd; it does NOT use the
LayerNorm-of-tokens / derived-query algebra of the HIP path.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/cae_*.npz; tests/test_cae_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class CaePort(nn.Module):
    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                     # cae_att.py:29
        self.query_token = nn.Parameter(torch.zeros(1, 1, dim))    # :86
        self.norm1_q, self.norm1_k, self.norm1_v, self.norm2_cross = (nn.LayerNorm(dim) for _ in range(4))   # :87-90
        self.q = nn.Linear(dim, dim, bias=False)                   # :31-33
        self.k = nn.Linear(dim, dim, bias=False)
        self.v = nn.Linear(dim, dim, bias=False)
        self.proj = nn.Linear(dim, dim)                            # :44

    def forward(self, x_kv, cls=None):
        B, N, C = x_kv.shape
        H = self.num_heads
        x_q = self.norm1_q(self.query_token.expand(B, -1, -1))     # :100-102
        x_k = self.norm1_k(x_kv)                                   # :103
        x_v = self.norm1_v(x_kv)                                   # :104
        q = self.q(x_q).reshape(B, 1, H, -1).permute(0, 2, 1, 3) * self.scale        # :59-60,68
        k = self.k(x_k).reshape(B, N, H, -1).permute(0, 2, 1, 3)                     # :62-63
        v = self.v(x_v).reshape(B, N, H, -1).permute(0, 2, 1, 3)                     # :65-66
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                             # :69-71
        x = (attn @ v).transpose(1, 2).reshape(B, 1, -1)                             # :74
        return self.proj(x).squeeze(1)                                               # :75,108


def make_head(dim, nb_classes, num_heads=8):
    return nn.Sequential(CaePort(dim, num_heads), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["query", "nq_w", "nq_b", "nk_w", "nk_b", "nv_w", "nv_b", "n2_w", "n2_b", "q_w", "k_w", "v_w", "proj_w", "proj_b",
               "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.query_token, p.norm1_q.weight, p.norm1_q.bias, p.norm1_k.weight, p.norm1_k.bias, p.norm1_v.weight,
            p.norm1_v.bias, p.norm2_cross.weight, p.norm2_cross.bias, p.q.weight, p.k.weight, p.v.weight, p.proj.weight,
            p.proj.bias, head[2].weight, head[2].bias]
