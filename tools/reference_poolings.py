"""Development aid (NOT part of the product package): factories for the REFERENCE's own pooling modules, to plug into
``efficient_probing_amd.probe_heads.register_pooling`` for side-by-side runs when the reference repository is importable
(``poolings.*`` on sys.path).  Every registry name is native in the package itself; nothing there imports this file.

    from tools.reference_poolings import reference_pooling
    probe_heads.register_pooling("dinovit", reference_pooling("dinovit"))

Constructor arguments follow reference probe_heads.py:42-84."""
from __future__ import annotations

import importlib

# how the reference builds the pooling modules we do not (yet) run natively:
# name -> (module path, class name, kwargs(dim, args, model))
REFERENCE_SPECS = {
    "abmilp": ("poolings.abmilp", "ABMILPHead",
               lambda dim, a, m: dict(dim=dim, self_attention_apply_to=a.abmilp_sa, activation=a.abmilp_act,
                                      depth=a.abmilp_depth, cond=a.abmilp_cond, content=a.abmilp_content,
                                      num_patches=m.patch_embed.num_patches)),
    "simpool": ("poolings.simpool", "SimPool",
                lambda dim, a, m: dict(dim=dim, num_heads=1, qkv_bias=False, qk_scale=None, gamma=None,
                                       use_beta=False)),
    "esimpool": ("poolings.simpool", "SimPool_nolinears",
                 lambda dim, a, m: dict(dim=dim, num_heads=12, qk_scale=None, gamma=None, use_beta=False)),
    "clip": ("poolings.clip.attention_pool2d", "AttentionPool2d",
             lambda dim, a, m: dict(in_features=dim, feat_size=16 if a.model == "capi_vitl14_in1k" else 14)),
    "siglip": ("poolings.clip.attention_pool", "AttentionPoolLatent", lambda dim, a, m: dict(in_features=dim)),
    "aim": ("poolings.aim", "AttentionPoolingClassifier", lambda dim, a, m: dict(dim=dim, num_heads=a.num_heads)),
    "cbam": ("poolings.cbam", "CbamPooling", lambda dim, a, m: dict(channels=dim, spatial_kernel_size=7)),
    "coca": ("poolings.coca_pytorch", "CrossAttention", lambda dim, a, m: dict(dim=dim)),
    "cait": ("poolings.other_pool", "CAPooling", lambda dim, a, m: dict(embed_dim=dim)),
    "dinovit": ("poolings.other_pool", "DinoViTBlockPooling", lambda dim, a, m: dict(d_model=dim)),
    "jepa": ("poolings.jepa.attentive_pooler", "AttentivePooler",
             lambda dim, a, m: dict(embed_dim=dim, num_heads=a.num_heads)),
    "dolg": ("poolings.dolg.dolg", "SpatialAttention2d",
             lambda dim, a, m: dict(in_c=dim, s3_dim=dim, with_aspp=False)),
    "cae": ("poolings.cae_att", "CAEAttentiveBlock", lambda dim, a, m: dict(dim=dim)),
}


def reference_pooling(name: str):
    mod_name, cls_name, kwargs = REFERENCE_SPECS[name]

    def make(dim, args, model):
        try:
            cls = getattr(importlib.import_module(mod_name), cls_name)
        except Exception as e:  # the reference repo is not on sys.path
            raise NotImplementedError(
                f"--cls_features {name}: the reference module "
                f"{mod_name}.{cls_name} is not importable ({e}); put the reference repository on sys.path or "
                f"register a factory with efficient_probing_amd.probe_heads.register_pooling().") from e
        return cls(**kwargs(dim, args, model))
    return make


