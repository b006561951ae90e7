"""AbMILP probe head (``--cls_features abmilp``), native on MI355X.

Same constructor, parameter names / shapes and initialisation ORDER as the reference ``ABMILPHead``
(reference poolings/abmilp.py:11-75) with its single-head ``Attention`` (reference models_vit.py:43-97), so a head built
under ``torch.manual_seed(s)`` has bit-identical initial weights and reference checkpoints (keys
``self_attn.qkv.weight``, ``self_attn.proj.{weight,bias}``, ``attention_predictor.{0,2}.{weight,bias}``) load with
``strict=True``.

Supported configuration = what the reference's command-line defaults build (main_linprobe.py:101-110):
self-attention applied to ``"both"``, ``tanh`` predictor of depth 2, no positional conditioning; ``content="patch"``
drops the first token like the reference.  Other options raise.  On a GPU the whole head runs in the HIP kernels
(csrc/ep_abmilp.hip); there is no CPU implementation here.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch import nn

from .. import functional as F_


class Attention(nn.Module):
    """Parameter container with the reference's names (models_vit.py:43-69): qkv without bias, proj with bias."""

    def __init__(self, dim: int, num_heads: int = 1, qkv_bias: bool = False):
        super().__init__()
        if num_heads != 1 or qkv_bias:
            raise NotImplementedError("native AbMILP: Attention(num_heads=1, qkv_bias=False) as built by ABMILPHead")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class ABMILPHead(nn.Module):
    def __init__(self, dim: int, self_attention_apply_to: str = "none", activation: str = "tanh", depth: int = 2,
                 cond: Optional[str] = "none", content: str = "all", num_patches: Optional[int] = None):
        super().__init__()
        if self_attention_apply_to != "both" or activation != "tanh" or depth != 2 or cond not in (None, "none"):
            raise NotImplementedError(
                "native AbMILP supports the reference's defaults (abmilp_sa='both', abmilp_act='tanh', abmilp_depth=2, "
                f"no abmilp_cond); got sa={self_attention_apply_to!r} act={activation!r} depth={depth} cond={cond!r}")
        if content not in ("all", "patch"):
            raise ValueError(f"content must be 'all' or 'patch', got {content!r}")
        self.cond = cond
        self.self_attention_apply_to = self_attention_apply_to
        self.content = content
        self.pos_embed = None
        self.self_attn = Attention(dim, num_heads=1)                      # abmilp.py:38
        self.ATTENTION_BRANCHES = 1
        self.attention_predictor = nn.Sequential(nn.Linear(dim, dim), nn.Tanh(), nn.Linear(dim, 1))   # abmilp.py:43-52

    def _tensors(self):
        p = self.attention_predictor
        return (self.self_attn.qkv.weight, self.self_attn.proj.weight, self.self_attn.proj.bias, p[0].weight, p[0].bias,
                p[2].weight, p[2].bias)

    def forward_with_attn_map(self, x: torch.Tensor):
        if x.dim() != 3 or x.shape[-1] != self.self_attn.qkv.in_features:
            raise ValueError(f"expected tokens (B, N, {self.self_attn.qkv.in_features}), got {tuple(x.shape)}")
        if self.content == "patch":
            x = x[:, 1:]                                                   # abmilp.py:56-57
        out_dtype = x.dtype
        out, amap = F_.abmilp_pool(x, *self._tensors())
        if out_dtype != torch.float32:
            out = out.to(out_dtype)
        return out, amap.unsqueeze(-1)                                     # (B, N, 1) like F.softmax(attn_map, dim=1)

    def forward(self, x: torch.Tensor, cls: Any = None, **_: Any) -> torch.Tensor:
        return self.forward_with_attn_map(x)[0]
