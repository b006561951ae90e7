"""CPU oracle for the CaiT class-attention probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features cait``: ``CAPooling(embed_dim)`` with one
``LayerScale_Block_CA`` / ``Class_Attention`` and the timm-style ``Mlp`` (reference poolings/other_pool.py:390-507,
poolings/clip/mlp.py:13-50) behind ``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's ``Linear`` (reference
probe_heads.py:79,105-106).  It keeps the reference's association -- concatenate the class token, LayerNorm all N + 1 rows,
q from the class row, k / v Linear over every row, per-head softmax over N + 1 entries, proj, LayerScale residuals, MLP,
final LayerNorm -- and gradients come from autograd; it does NOT use the derived-query / merged-entry algebra of the HIP path.

PARITY PIN: golden vectors produced by importing the real reference module (tests/golden/make_golden.py ->
tests/golden/cait_*.npz; tests/test_cait_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.
"""
from __future__ import annotations

import torch
from torch import nn


class CaitPort(nn.Module):
    def __init__(self, dim, num_heads=4, mlp_ratio=4.0, init_scale=1e-5):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5                    # other_pool.py:446-447
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)                   # :487 (norm_layer = LayerNorm eps 1e-6, :395)
        self.q = nn.Linear(dim, dim)                               # :449-451 (qkv_bias=True, :394)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        self.proj = nn.Linear(dim, dim)                            # :453
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)                   # :491
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))            # clip/mlp.py:34
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)            # clip/mlp.py:38
        self.gamma_1 = nn.Parameter(init_scale * torch.ones(dim))  # :494-495
        self.gamma_2 = nn.Parameter(init_scale * torch.ones(dim))
        self.norm = nn.LayerNorm(dim)                              # :415
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))      # :416

    def forward(self, x, cls=None):
        B, N, C = x.shape
        H = self.num_heads
        x_cls = self.cls_token.expand(B, -1, -1)                                            # :428
        u = self.norm1(torch.cat((x_cls, x), dim=1))                                        # :500,503
        q = self.q(u[:, 0]).unsqueeze(1).reshape(B, 1, H, C // H).permute(0, 2, 1, 3) * self.scale     # :458,461
        k = self.k(u).reshape(B, N + 1, H, C // H).permute(0, 2, 1, 3)                     # :459
        v = self.v(u).reshape(B, N + 1, H, C // H).permute(0, 2, 1, 3)                     # :462
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                                    # :464-465
        a = self.proj((attn @ v).transpose(1, 2).reshape(B, 1, C))                          # :468-469
        x_cls = x_cls + self.gamma_1 * a                                                    # :503
        x_cls = x_cls + self.gamma_2 * self.fc2(torch.nn.functional.gelu(self.fc1(self.norm2(x_cls))))   # :505
        return self.norm(torch.cat((x_cls, x), dim=1))[:, 0]                                # :433-436


def make_head(dim, nb_classes, num_heads=4):
    return nn.Sequential(CaitPort(dim, num_heads), nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["cls_token", "gamma_1", "gamma_2", "n1_w", "n1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "proj_w", "proj_b",
               "n2_w", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "norm_w", "norm_b", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.cls_token, p.gamma_1, p.gamma_2, p.norm1.weight, p.norm1.bias, p.q.weight, p.q.bias, p.k.weight, p.k.bias,
            p.v.weight, p.v.bias, p.proj.weight, p.proj.bias, p.norm2.weight, p.norm2.bias, p.fc1.weight, p.fc1.bias,
            p.fc2.weight, p.fc2.bias, p.norm.weight, p.norm.bias, head[2].weight, head[2].bias]
