"""CPU oracle for the AbMILP probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features abmilp`` at its command-line defaults
(reference main_linprobe.py:101-110): ``ABMILPHead(dim, self_attention_apply_to="both", activation="tanh",
depth=2)`` (reference poolings/abmilp.py:11-75) with the single-head ``Attention`` of reference models_vit.py:43-97,
behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's ``Linear`` (reference probe_heads.py:42-51,67,105-106).
Gradients come from autograd on this graph.

PARITY PIN: checked against golden vectors produced by importing the real reference modules
(tests/golden/make_golden.py -> tests/golden/abmilp_*.npz; tests/test_abmilp_cpu.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this file;
the product package never does.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class AttentionPort(nn.Module):
    def __init__(self, dim, num_heads=1):                       # models_vit.py:46-69
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):                                       # models_vit.py:71-96
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q = q * self.scale
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x), attn


class AbmilpPort(nn.Module):
    def __init__(self, dim, content="all"):
        super().__init__()
        self.content = content
        self.self_attn = AttentionPort(dim, num_heads=1)        # abmilp.py:38
        self.attention_predictor = nn.Sequential(nn.Linear(dim, dim), nn.Tanh(), nn.Linear(dim, 1))   # :43-52

    def forward_with_attn_map(self, x):                         # abmilp.py:54-67
        if self.content == "patch":
            x = x[:, 1:]
        x_attn, _ = self.self_attn(x)
        attn_map = F.softmax(self.attention_predictor(x_attn), dim=1)
        return (x_attn * attn_map).sum(dim=1), attn_map

    def forward(self, x, cls=None):
        return self.forward_with_attn_map(x)[0]


def make_head(dim, nb_classes, content="all"):
    return nn.Sequential(AbmilpPort(dim, content), nn.BatchNorm1d(dim, affine=False, eps=1e-6),
                         nn.Linear(dim, nb_classes))


PARAM_NAMES = ["qkv", "proj_w", "proj_b", "w1", "b1", "w2", "b2", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.self_attn.qkv.weight, p.self_attn.proj.weight, p.self_attn.proj.bias, p.attention_predictor[0].weight,
            p.attention_predictor[0].bias, p.attention_predictor[2].weight, p.attention_predictor[2].bias,
            head[2].weight, head[2].bias]
