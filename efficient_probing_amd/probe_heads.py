"""Probe-head registry: ``--cls_features`` name -> how to build the pooling module and which
classifier sits behind it.  Call surface of reference probe_heads.py:60-110:

    POOLINGS[name] = (make_pooling(dim, args, model) -> nn.Module,
                      make_classifier(dim, args) -> nn.Linear | None)
    build_probe_head(model, args)   # model.head <- Sequential(pooling, BatchNorm1d, classifier)

Invariants kept from the reference (published numbers depend on them):
  * the pooling module is constructed BEFORE the classifier, so the RNG draws happen in the
    same order under a fixed seed (reference probe_heads.py:14-16,104);
  * only EP builds a fresh classifier (its width is dim // d_out); every other pooling keeps the
    encoder's own ``model.head`` object, which the --finetune checkpoint already wrote into
    (reference probe_heads.py:17-20,105);
  * a ``_all`` suffix selects the same pooling over [CLS]+patch tokens (reference :95);
  * names without an entry (cls, gap, raw, both, ...) get BatchNorm + the encoder's head.

Native on MI355X: ``ep``, ``coca``, ``abmilp``, ``siglip``, ``cae``, ``jepa``, ``aim``, ``simpool``, ``esimpool``, ``cait``,
``clip``, ``dolg``, ``cbam`` and ``dinovit`` -- all fourteen names (pooling, BatchNorm1d and the classifier run in the HIP kernels
of libep_hip.so).  ``register_pooling`` plugs in any other factory (``tools/reference_poolings.py`` builds factories for the
reference's own PyTorch modules when its repository is importable -- a development aid kept OUTSIDE this package); such
modules run as stock PyTorch-ROCm modules behind the native BatchNorm.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import functional as F_
from .poolings.ep import EfficientProbing
from .poolings.coca import CrossAttention as CocaPooling
from .poolings.abmilp import ABMILPHead
from .poolings.siglip import AttentionPoolLatent
from .poolings.cae import CAEAttentiveBlock
from .poolings.jepa import AttentivePooler
from .poolings.aim import AttentionPoolingClassifier
from .poolings.simpool import SimPool, SimPool_nolinears
from .poolings.cait import CAPooling
from .poolings.clip import AttentionPool2d
from .poolings.dinovit import DinoViTBlockPooling
from .poolings.dolg import SpatialAttention2d
from .poolings.cbam import CbamPooling
from .util.cls_features import ATTENTIVE_POOLINGS, base_pooling_name

BN_EPS = 1e-6


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d(width, affine=False, eps=1e-6) whose GPU fp32 path is the native kernel
    (batch statistics per GPU, running stats with momentum 0.1 and unbiased variance; same
    buffers / state-dict keys).  The class keeps the torch name so ``repr(head)`` is unchanged."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        native = x.is_cuda and x.dim() == 2 and not self.affine and self.track_running_stats \
            and self.momentum is not None
        if not native:
            return super().forward(x)
        out_dtype = x.dtype
        if self.training:
            z = F_.batch_norm_train(x, self.running_mean, self.running_var, self.num_batches_tracked,
                                    self.eps, self.momentum)
        else:
            z = F_.bn_forward_eval(x, self.running_mean, self.running_var, self.eps)
        return z if out_dtype == torch.float32 else z.to(out_dtype)


class Linear(nn.Linear):
    """nn.Linear whose GPU path is the native f32-MFMA kernel (same init, same state dict)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not (x.is_cuda and x.dim() == 2):
            return super().forward(x)
        out_dtype = x.dtype
        y = F_.linear(x, self.weight, self.bias)
        return y if out_dtype == torch.float32 else y.to(out_dtype)


def _batchnorm(width: int) -> nn.Module:
    return BatchNorm1d(width, affine=False, eps=BN_EPS)


PoolingFactory = Callable[[int, object, nn.Module], nn.Module]
ClassifierFactory = Optional[Callable[[int, object], nn.Linear]]

def _make_ep(dim, args, model):
    return EfficientProbing(dim=dim, num_queries=args.ep_queries, d_out=args.d_out)


def _make_ep_classifier(dim, args):
    return Linear(dim // args.d_out, args.nb_classes, bias=True)


POOLINGS: Dict[str, Tuple[PoolingFactory, ClassifierFactory]] = {}
POOLINGS["ep"] = (_make_ep, _make_ep_classifier)
POOLINGS["abmilp"] = (lambda dim, args, model: ABMILPHead(       # native (reference probe_heads.py:42-51)
    dim=dim, self_attention_apply_to=args.abmilp_sa, activation=args.abmilp_act, depth=args.abmilp_depth,
    cond=args.abmilp_cond, content=args.abmilp_content, num_patches=model.patch_embed.num_patches), None)
POOLINGS["siglip"] = (lambda dim, args, model: AttentionPoolLatent(in_features=dim), None)   # native (:72)
POOLINGS["cae"] = (lambda dim, args, model: CAEAttentiveBlock(dim=dim), None)               # native (:83)
POOLINGS["jepa"] = (lambda dim, args, model: AttentivePooler(embed_dim=dim, num_heads=args.num_heads), None)   # native (:81)
POOLINGS["coca"] = (lambda dim, args, model: CocaPooling(dim=dim), None)     # native (reference probe_heads.py:78)
POOLINGS["simpool"] = (lambda dim, args, model: SimPool(dim=dim, num_heads=1, qkv_bias=False, qk_scale=None, gamma=None,
                                                        use_beta=False), None)                 # native (:66-67)
POOLINGS["esimpool"] = (lambda dim, args, model: SimPool_nolinears(dim=dim, num_heads=12, qk_scale=None, gamma=None,
                                                                   use_beta=False), None)      # native (:68-69)
POOLINGS["cait"] = (lambda dim, args, model: CAPooling(embed_dim=dim), None)                    # native (:79)
POOLINGS["clip"] = (lambda dim, args, model: AttentionPool2d(                                   # native (:54-57,71)
    in_features=dim, feat_size=16 if getattr(args, "model", None) == "capi_vitl14_in1k" else 14), None)
POOLINGS["dolg"] = (lambda dim, args, model: SpatialAttention2d(in_c=dim, s3_dim=dim, with_aspp=False), None)   # native (:82)
POOLINGS["cbam"] = (lambda dim, args, model: CbamPooling(channels=dim, spatial_kernel_size=7), None)   # native (:77)
POOLINGS["dinovit"] = (lambda dim, args, model: DinoViTBlockPooling(d_model=dim), None)          # native (:80)
POOLINGS["aim"] = (lambda dim, args, model: AttentionPoolingClassifier(dim=dim, num_heads=args.num_heads), None)   # native (:73)


def register_pooling(name: str, make_pooling: PoolingFactory, make_classifier: ClassifierFactory = None) -> None:
    """Plug another pooling (or a native re-implementation) into the registry."""
    if name not in ATTENTIVE_POOLINGS:
        raise KeyError(f"{name!r} is not an attentive pooling known to map_cls_features()")
    POOLINGS[name] = (make_pooling, make_classifier)


def build_probe_head(model: nn.Module, args) -> None:
    """Replace ``model.head`` in place with the probe selected by ``args.cls_features``.
    Must run after the --finetune checkpoint load and before --resume
    (reference probe_heads.py:87-106, main_linprobe.py:496-500)."""
    base = base_pooling_name(args.cls_features)
    dim = model.head.in_features
    if base not in POOLINGS:
        model.head = nn.Sequential(_batchnorm(dim), model.head)       # plain linear probing
        return
    make_pooling, make_classifier = POOLINGS[base]
    pooling = make_pooling(dim, args, model)                           # first: fixes the RNG order
    classifier = make_classifier(dim, args) if make_classifier is not None else model.head
    model.head = nn.Sequential(pooling, _batchnorm(classifier.in_features), classifier)


# pooling class -> (name of its fused engine class in engine.py, short kind).  ONE table instead of a predicate per head:
# ``native_head_kind`` names the kind of a Sequential(pooling, BatchNorm1d, Linear) built by this registry (or "lp" for
# plain linear probing), ``engine.make_engine`` looks the engine class up here.
NATIVE_HEADS = {
    EfficientProbing: ("ProbeHeadEngine", "ep"),
    CocaPooling: ("CocaHeadEngine", "coca"),
    ABMILPHead: ("AbmilpHeadEngine", "abmilp"),
    AttentionPoolLatent: ("SiglipHeadEngine", "siglip"),
    CAEAttentiveBlock: ("CaeHeadEngine", "cae"),
    AttentivePooler: ("JepaHeadEngine", "jepa"),
    AttentionPoolingClassifier: ("AimHeadEngine", "aim"),
    SimPool: ("SimpoolHeadEngine", "simpool"),
    SimPool_nolinears: ("SimpoolHeadEngine", "simpool"),
    CAPooling: ("CaitHeadEngine", "cait"),
    AttentionPool2d: ("ClipHeadEngine", "clip"),
    SpatialAttention2d: ("DolgHeadEngine", "dolg"),
    CbamPooling: ("CbamHeadEngine", "cbam"),
    DinoViTBlockPooling: ("DinovitHeadEngine", "dinovit"),
}


def native_head_kind(head: nn.Module):
    """"ep", "coca", ... for Sequential(<native pooling>, BatchNorm1d, Linear); "lp" for Sequential(BatchNorm1d(affine=False),
    Linear) (plain linear probing, reference probe_heads.py:96-99); None for anything else."""
    if not isinstance(head, nn.Sequential):
        return None
    if len(head) == 2 and isinstance(head[0], nn.BatchNorm1d) and not head[0].affine and isinstance(head[1], nn.Linear):
        return "lp"
    if len(head) == 3 and isinstance(head[1], nn.BatchNorm1d) and isinstance(head[2], nn.Linear):
        entry = _native_entry(head[0])
        return entry[1] if entry else None
    return None


def _native_entry(pool: nn.Module):
    """NATIVE_HEADS entry of a pooling module, resolved along its MRO: a user subclass (``class MyEP(EfficientProbing)``)
    stays on the fused path of its base class instead of silently dropping to the module path."""
    return next((NATIVE_HEADS[c] for c in type(pool).__mro__ if c in NATIVE_HEADS), None)


def native_engine_name(head: nn.Module):
    """Name of the engine.py class that runs this head's fused step (None: not a native head)."""
    kind = native_head_kind(head)
    if kind == "lp":
        return "LinearProbeEngine"
    return _native_entry(head[0])[0] if kind else None


def is_native_head(head: nn.Module) -> bool:
    return native_head_kind(head) is not None


def __getattr__(name: str):
    """``is_native_<kind>_head(head)`` for every kind of the table (kept as the per-head spelling the engines' constructor
    checks and the tests use)."""
    m = __import__("re").fullmatch(r"is_native_([a-z0-9]+)_head", name)
    kinds = {k for _, k in NATIVE_HEADS.values()} | {"lp"}
    if m and m.group(1) in kinds:
        kind = m.group(1)
        return lambda head: native_head_kind(head) == kind
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


assert sorted(POOLINGS) == sorted(ATTENTIVE_POOLINGS), sorted(set(POOLINGS) ^ set(ATTENTIVE_POOLINGS))
