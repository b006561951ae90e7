"""Which tokens an encoder hands to the probe for a given ``--cls_features`` value
(behaviour of reference util/cls_features.py:19-37).

    "pos"                  -> "gap"   (mean over patch tokens, no pooling module)
    "<pooling>"            -> "pos"   (patch tokens only)
    "<pooling>_all"        -> "both"  ([CLS] + patch tokens)
    anything else          -> itself
"""

ATTENTIVE_POOLINGS = (
    "abmilp", "simpool", "esimpool", "clip", "siglip", "aim", "ep", "cbam", "coca",
    "cait", "dinovit", "jepa", "dolg", "cae",
)
ATTENTIVE_POOLINGS_ALL = tuple(f"{p}_all" for p in ATTENTIVE_POOLINGS)

_SELECTION = {"pos": "gap"}
_SELECTION.update({p: "pos" for p in ATTENTIVE_POOLINGS})
_SELECTION.update({p: "both" for p in ATTENTIVE_POOLINGS_ALL})


def map_cls_features(return_features):
    return _SELECTION.get(return_features, return_features)


def base_pooling_name(cls_features: str) -> str:
    """'ep_all' -> 'ep' (reference probe_heads.py:95)."""
    return cls_features[:-4] if cls_features.endswith("_all") else cls_features
