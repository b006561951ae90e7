"""CPU-only checks: the C-ABI library loads and exports what include/ep_hip.h declares, and the
host-side mirror of the reference interface (registry, schedule, scaler, optimizer bookkeeping)
behaves like the reference.  No kernel is launched here."""
import hashlib
import json
import os
import re
from argparse import Namespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FX = json.load(open(os.path.join(GOLD, "host_fixtures.json")))


def test_library_exports_every_declared_symbol():
    from efficient_probing_amd import _native
    header = open(os.path.join(ROOT, "include", "ep_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ep_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = _native.load()                       # resolves every symbol in _native.SIGNATURES
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    # one ABI version everywhere: the header, the library and the ctypes structs (load() raises on a stale .so)
    abi = int(re.search(r"#define\s+EP_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.ep_version() == abi == _native.EP_ABI_VERSION
    assert isinstance(_native.last_error(), str)


def test_stale_library_is_rejected(monkeypatch):
    """a git-ignored .so of another ABI version still exports every symbol and would misread the step
    structs; load() must refuse it."""
    from efficient_probing_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "EP_ABI_VERSION", _native.EP_ABI_VERSION + 1)
    with pytest.raises(_native.NativeLibraryError, match="ABI version"):
        _native.load()


def test_param_layout_matches_parameters_order():
    from efficient_probing_amd.parallel import head_param_layout, head_param_shapes
    offs, total = head_param_layout(768, 8, 1, 1000)
    assert offs == [0, 8 * 768, 8 * 768 + 768 * 768, 8 * 768 + 768 * 768 + 1000 * 768]
    assert total == offs[3] + 1000
    offs, total = head_param_layout(64, 4, 2, 10)        # bias of 10 is padded to 12
    assert offs[3] + 12 == total and all(o % 4 == 0 for o in offs)
    assert head_param_shapes(64, 4, 2, 10) == [(1, 4, 64), (32, 64), (10, 32), (10,)]


def test_workspace_queries_fail_cleanly_on_bad_dims():
    import ctypes as C
    from efficient_probing_amd import _native
    lib = _native.load()
    bad = _native.EPHeadDims(B=4, N=16, D=64, Q=5, d_out=1, C=10)     # 64 % 5 != 0
    assert lib.ep_head_workspace_bytes(C.byref(bad)) == 0
    assert "d_out*Q" in _native.last_error()
    ok = _native.EPHeadDims(B=4, N=16, D=64, Q=4, d_out=1, C=10)
    assert lib.ep_head_workspace_bytes(C.byref(ok)) > 0


class StubEncoder(torch.nn.Module):
    def __init__(self, dim, classes):
        super().__init__()
        self.embed_dim = dim
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, classes)


def _args(**kw):
    a = Namespace(cls_features="ep", ep_queries=32, d_out=1, nb_classes=1000, num_heads=16, abmilp_sa="both",
                  abmilp_act="tanh", abmilp_depth=2, abmilp_cond=None, abmilp_content="all", model="vit_base_patch16")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("key", sorted(FX["init"]))
def test_head_init_is_bit_identical_to_the_reference(key):
    """Same modules created in the same order under the same seed -> same RNG draws
    (reference probe_heads.py:14-16; fingerprint method of tools/inv_heads.py:102-120)."""
    from efficient_probing_amd import probe_heads
    m = re.match(r"d(\d+)_q(\d+)_o(\d+)_c(\d+)", key)
    dim, Q, d_out, C_ = map(int, m.groups())
    want = FX["init"][key]
    torch.manual_seed(0)
    enc = StubEncoder(dim, C_)
    probe_heads.build_probe_head(enc, _args(ep_queries=Q, d_out=d_out, nb_classes=C_))
    head = enc.head
    assert repr(head) == want["repr"]
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == want["keys"]
    for k, v in sd.items():
        assert hashlib.sha256(v.detach().contiguous().numpy().tobytes()).hexdigest() == want["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == want["n_trainable"]
    # published head size: D^2/d_out + Q*D + C*D/d_out + C   (reference tools/gen_leaderboard.py:470-471 at d_out=1)
    assert want["n_trainable"] == dim * dim // d_out + Q * dim + C_ * dim // d_out + C_


def test_registry_surface():
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.util.cls_features import ATTENTIVE_POOLINGS, map_cls_features, base_pooling_name
    assert sorted(probe_heads.POOLINGS) == sorted(ATTENTIVE_POOLINGS) and len(ATTENTIVE_POOLINGS) == 14
    assert map_cls_features("ep") == "pos" and map_cls_features("ep_all") == "both"
    assert map_cls_features("pos") == "gap" and map_cls_features("cls") == "cls"
    assert base_pooling_name("ep_all") == "ep" and base_pooling_name("coca") == "coca"
    # `_all` selects the same pooling; EP builds a fresh classifier even at d_out = 1
    enc = StubEncoder(64, 10)
    own = enc.head
    probe_heads.build_probe_head(enc, _args(cls_features="ep_all", ep_queries=4, nb_classes=10))
    assert probe_heads.is_native_ep_head(enc.head) and enc.head[2] is not own
    assert enc.head[1].eps == 1e-6 and enc.head[1].affine is False
    # plain linear probing keeps the encoder's own Linear object (IDENTITY property)
    enc = StubEncoder(64, 10)
    own = enc.head
    probe_heads.build_probe_head(enc, _args(cls_features="cls"))
    assert len(enc.head) == 2 and enc.head[1] is own
    # every registry name is native and the product package holds no hook that imports the reference's modules; the
    # development aid tools/reference_poolings.py fails loudly when the reference repository is not on sys.path
    assert all(probe_heads.POOLINGS[n][0] is not None for n in ATTENTIVE_POOLINGS)
    enc = StubEncoder(64, 10)
    probe_heads.build_probe_head(enc, _args(cls_features="dinovit"))
    assert probe_heads.is_native_dinovit_head(enc.head) and probe_heads.is_native_head(enc.head)
    assert not hasattr(probe_heads, "_reference_pooling") and "importlib" not in vars(probe_heads)
    from tools.reference_poolings import reference_pooling
    with pytest.raises(NotImplementedError, match="register_pooling"):
        reference_pooling("dinovit")(64, _args(cls_features="dinovit"), enc)
    # ... another implementation can be plugged in; it keeps the encoder's classifier
    native = probe_heads.POOLINGS["dinovit"]
    probe_heads.register_pooling("dinovit", lambda dim, a, m: torch.nn.Identity())
    try:
        enc = StubEncoder(64, 10)
        own = enc.head
        probe_heads.build_probe_head(enc, _args(cls_features="dinovit"))
        assert isinstance(enc.head[0], torch.nn.Identity) and enc.head[2] is own
    finally:
        probe_heads.POOLINGS["dinovit"] = native
    with pytest.raises(KeyError):
        probe_heads.register_pooling("nonsense", lambda *a: None)


def test_constructor_validation_and_no_cpu_fallback():
    from efficient_probing_amd.poolings.ep import EfficientProbing
    with pytest.raises(ValueError):
        EfficientProbing(dim=64, num_queries=5)
    with pytest.raises(NotImplementedError):
        EfficientProbing(dim=64, num_heads=2, num_queries=4)
    m = EfficientProbing(dim=64, num_queries=4, d_out=2)
    assert m.v.weight.shape == (32, 64) and m.cls_token.shape == (1, 4, 64) and m.scale == 64 ** -0.5
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.randn(2, 5, 64))                 # CPU tokens: the product path must not silently fall back
    from efficient_probing_amd.util.lars import LARS
    p = torch.nn.Parameter(torch.randn(3, 3))
    p.grad = torch.randn(3, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        LARS([p], lr=0.1).step()
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine
    enc = StubEncoder(64, 10)
    probe_heads.build_probe_head(enc, _args(ep_queries=4, nb_classes=10))
    with pytest.raises(RuntimeError, match="GPU"):
        ProbeHeadEngine(enc.head)
    # the arithmetic switch (ep_head_step.arith): two names, checked before anything touches a device
    with pytest.raises(ValueError, match="arithmetic"):
        ProbeHeadEngine(enc.head, arithmetic="fp8")
    with pytest.raises(RuntimeError, match="GPU"):
        ProbeHeadEngine(enc.head, arithmetic="bf16_autocast")
    from efficient_probing_amd import _native
    assert (_native.EP_ARITH_F32, _native.EP_ARITH_BF16_AUTOCAST) == (0, 1)
    names = [n for n, _ in _native.EPHeadStep._fields_]
    assert names[-7:-5] == ["planes_valid", "arith"]                                          # ABI v25: appended, nothing moved
    assert names[-5:] == ["scaler_state", "scaler_slot", "scaler_growth", "scaler_backoff", "scaler_interval"]   # ABI v26: likewise


def test_lr_schedule_matches_reference_table():
    from efficient_probing_amd.util.lr_sched import adjust_learning_rate, absolute_lr
    for row in FX["lr"]:
        opt = Namespace(param_groups=[{"lr": -1.0}, {"lr": -1.0, "lr_scale": 0.5}])
        got = adjust_learning_rate(opt, row["epoch"], Namespace(lr=row["lr"], min_lr=row["min_lr"],
                                                                warmup_epochs=row["warmup"], epochs=row["epochs"]))
        assert got == pytest.approx(row["out"], rel=1e-12, abs=1e-15)
        assert opt.param_groups[0]["lr"] == pytest.approx(row["group0"], rel=1e-12, abs=1e-15)
        assert opt.param_groups[1]["lr"] == pytest.approx(row["group1"], rel=1e-12, abs=1e-15)
    assert absolute_lr(0.1, 4096) == pytest.approx(1.6)


def test_loss_scaler_follows_gradscaler_trajectory():
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    fx = FX["scaler"]
    sc = NativeScalerWithGradNormCount(growth_interval=fx["growth_interval"])
    assert sc.state_dict_key == "amp_scaler" and sc.get_scale() == 65536.0
    for i, want in enumerate(fx["scale"]):
        sc.update(i in fx["inf_at"])
        assert sc.get_scale() == want
    st = sc.state_dict()
    sc2 = NativeScalerWithGradNormCount()
    sc2.load_state_dict(st)
    assert sc2.get_scale() == sc.get_scale()
    # generic (non-native) optimizer path on CPU: unscale, norm, step
    p = torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.SGD([p], lr=0.1)
    sc3 = NativeScalerWithGradNormCount()
    norm = sc3((p * torch.tensor([1.0, 2.0, 2.0])).sum(), opt, parameters=[p])
    assert float(norm) == pytest.approx(3.0)
    assert torch.allclose(p.detach(), torch.tensor([0.9, 0.8, 0.8]))


def test_shard_range():
    from efficient_probing_amd.parallel import shard_range
    assert shard_range(10, 2, 0) == (0, 5) and shard_range(10, 2, 1) == (5, 10)
    assert shard_range(11, 4, 3) == (6, 8)


def test_subclass_of_a_native_pooling_keeps_the_fused_engine():
    """Round 4 advisor finding: the class -> engine table was looked up by exact type, so ``class MyEP(EfficientProbing)`` fell
    off the fused path silently.  The lookup walks the MRO now."""
    import torch
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.ep import EfficientProbing

    class MyEP(EfficientProbing):
        pass

    head = torch.nn.Sequential(MyEP(64, num_queries=4), torch.nn.BatchNorm1d(64, affine=False, eps=1e-6), torch.nn.Linear(64, 10))
    assert probe_heads.native_head_kind(head) == "ep"
    assert probe_heads.native_engine_name(head) == "ProbeHeadEngine"
    assert probe_heads.is_native_ep_head(head)
    plain = torch.nn.Sequential(torch.nn.Identity(), torch.nn.BatchNorm1d(64, affine=False), torch.nn.Linear(64, 10))
    assert probe_heads.native_head_kind(plain) is None
