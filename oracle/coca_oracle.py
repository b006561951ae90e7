"""CPU oracle for the CoCa attentional-pooler probe head -- TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of what the reference executes for ``--cls_features coca``:
``CrossAttention(dim)`` (reference poolings/coca_pytorch.py:250-343, LayerNorm :70-77) behind
``BatchNorm1d(affine=False, eps=1e-6)`` and the encoder's ``Linear`` (reference probe_heads.py:78,105-106).
It keeps the reference's association -- LayerNorm and ``to_q`` over ALL image queries, ``to_kv`` over every
token, ``sim - amax``, softmax, ``attn @ v``, ``to_out``, then ``[:, 0]`` -- and gradients come from autograd
on that graph.  It does NOT use the derived-query / pool-then-project algebra of the HIP path
(csrc/ep_coca.hip), so agreement between the two is a real check of that algebra.

PARITY PIN: checked against golden vectors produced by importing the real reference module
(tests/golden/make_golden.py -> tests/golden/coca_*.npz; tests/test_oracle_golden.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this file;
the product package never does.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class LayerNormPort(nn.Module):
    def __init__(self, dim):                                   # coca_pytorch.py:70-77
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x):
        return F.layer_norm(x, x.shape[-1:], self.gamma, self.beta)


class CocaPort(nn.Module):
    def __init__(self, dim, dim_head=64, num_img_queries=196, heads=8):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5                          # :266
        inner = heads * dim_head
        self.norm = LayerNormPort(dim)                         # :271
        self.context_norm = nn.Identity()                      # :272 (norm_context=False)
        self.img_queries = nn.Parameter(torch.randn(num_img_queries, dim))   # :274
        self.to_q = nn.Linear(dim, inner, bias=False)          # :276
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)  # :277
        self.to_out = nn.Linear(inner, dim, bias=False)        # :278

    def attention(self, context):
        B = context.shape[0]
        x = self.img_queries.unsqueeze(0).expand(B, -1, -1)    # :300-303
        x = self.norm(x)                                       # :307
        q = self.to_q(x)                                       # :312
        q = q.reshape(B, q.shape[1], self.heads, self.dim_head).permute(0, 2, 1, 3) * self.scale   # :313-317
        k, v = self.to_kv(context).chunk(2, dim=-1)            # :321
        sim = torch.einsum("bhid,bjd->bhij", q, k)             # :325
        sim = sim - sim.amax(dim=-1, keepdim=True)             # :329
        return sim.softmax(dim=-1), v                          # :330

    def forward(self, context, cls=None):
        attn, v = self.attention(context)
        out = torch.einsum("bhij,bjd->bhid", attn, v)          # :334
        out = out.permute(0, 2, 1, 3).reshape(out.shape[0], out.shape[2], -1)   # :338
        out = self.to_out(out)                                 # :339
        return out[:, 0]                                       # :343


def make_head(dim, nb_classes, **kw):
    """Sequential(pooling, BN, classifier) as reference probe_heads.py:102-106 builds it for 'coca'."""
    pool = CocaPort(dim, **kw)
    return nn.Sequential(pool, nn.BatchNorm1d(dim, affine=False, eps=1e-6), nn.Linear(dim, nb_classes))


PARAM_NAMES = ["gamma", "img_queries", "to_q", "to_kv", "to_out", "fc_weight", "fc_bias"]


def head_params(head):
    p = head[0]
    return [p.norm.gamma, p.img_queries, p.to_q.weight, p.to_kv.weight, p.to_out.weight, head[2].weight, head[2].bias]
