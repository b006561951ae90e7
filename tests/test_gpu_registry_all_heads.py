"""Every ``--cls_features`` name of the registry end to end on the GPU, the way main_linprobe drives it (reference
main_linprobe.py:496-500, 581-659): ``build_probe_head`` on an encoder stub -> ``train_one_epoch`` (fused engine) ->
``evaluate`` -> ``save_model`` -> a freshly built head -> ``load_model`` (strict) -> the same evaluation.  The per-head parity
tests pin the numbers; this one pins the plumbing for all fourteen names at once.  Needs an MI355X (pytest -m gpu)."""
import math
from argparse import Namespace

import pytest
import torch

from efficient_probing_amd.util.cls_features import ATTENTIVE_POOLINGS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D, N_TOK, C, B = 384, 16, 12, 8                     # 12 x 32 channels (esimpool), 4 x 4 token grid (clip / dolg / cbam need a square one)


class Enc(torch.nn.Module):
    """The encoder's two attributes the registry reads (probe_heads.py:50,105) and the call the backbones make."""

    def __init__(self):
        super().__init__()
        self.patch_embed = Namespace(num_patches=N_TOK)
        self.head = torch.nn.Linear(D, C)

    def forward(self, tokens):
        return self.head(tokens)


def reg_args(name):
    return Namespace(cls_features=name, ep_queries=4, d_out=1, nb_classes=C, num_heads=4, abmilp_sa="both", abmilp_act="tanh",
                     abmilp_depth=2, abmilp_cond=None, abmilp_content="all", model="vit_base_patch16")


def make_model(name, seed=0):
    from efficient_probing_amd import probe_heads
    torch.manual_seed(seed)
    m = Enc()
    probe_heads.build_probe_head(m, reg_args(name))
    assert probe_heads.is_native_head(m.head), name
    return m.to(DEV)


def loader(n_batches, seed=5):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, N_TOK, D, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(n_batches)]


def fix_grid(m):
    """clip's position embedding is sized by the registry for a 14 x 14 grid: rebuild it for the 4 x 4 test grid."""
    from efficient_probing_amd.poolings.clip import AttentionPool2d
    if isinstance(m.head[0], AttentionPool2d):
        torch.manual_seed(0)
        m.head[0] = AttentionPool2d(in_features=D, feat_size=int(math.isqrt(N_TOK))).to(DEV)
    return m


@pytest.mark.parametrize("name", sorted(ATTENTIVE_POOLINGS))
def test_train_evaluate_checkpoint_round_trip(name, tmp_path):
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd import checkpoint as CK
    from efficient_probing_amd.engine import make_engine
    from efficient_probing_amd.util.lars import LARS
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    model = fix_grid(make_model(name))
    opt = LARS(model.head.parameters(), lr=0.0, weight_decay=0.0)
    args = Namespace(accum_iter=1, amp="none", lr=0.2, min_lr=0.0, warmup_epochs=0, epochs=2, output_dir=str(tmp_path), suffix="t",
                     resume="")
    before = [p.detach().clone() for p in model.head.parameters()]
    scaler = NativeScalerWithGradNormCount()
    stats = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader(3), opt, torch.device(DEV), 0, scaler, args=args)
    assert math.isfinite(stats["loss"]) and 0.0 <= stats["acc1"] <= 100.0
    assert EF.get_engine(model) is not None and type(EF.get_engine(model)) is type(make_engine(fix_grid(make_model(name)).head))
    moved = [not torch.equal(a, b.detach()) for a, b in zip(before, model.head.parameters())]
    assert any(moved), f"{name}: no parameter changed"
    assert int(model.head[1].num_batches_tracked) == 3
    ev = EF.evaluate(loader(2, seed=9), model, torch.device(DEV))
    assert math.isfinite(ev["loss"])
    # checkpoint of the head, resumed into a freshly built (differently initialised) head: identical evaluation
    path = CK.save_model(args, 0, model, model.head, opt, scaler, ev)
    other = fix_grid(make_model(name, seed=123))
    assert any(not torch.equal(a.detach(), b.detach()) for a, b in zip(other.head.parameters(), model.head.parameters()))
    rargs = Namespace(**{**vars(args), "resume": str(path)})
    CK.load_model(rargs, other, optimizer=None, loss_scaler=None, strict=True)
    for (k, a), (_, b) in zip(model.head.state_dict().items(), other.head.state_dict().items()):
        assert torch.equal(a, b), f"{name}: {k}"
    ev2 = EF.evaluate(loader(2, seed=9), other, torch.device(DEV))
    assert ev2["loss"] == ev["loss"] and ev2["acc1"] == ev["acc1"]


@pytest.mark.parametrize("name", sorted(ATTENTIVE_POOLINGS))
def test_resident_store_epoch_equals_gathered_batches(name):
    """The protocol's real input for every head: batches read IN PLACE from a resident token store
    (``ResidentTokenStore.loader`` -> ``StoreBatch``), through ``train_one_epoch`` and ``evaluate``.  The heads whose token passes
    take per-token LayerNorm statistics / per-image channel statistics look them up in the store's cached tables
    (``ResidentTokenStore.table``, attached by ``engine_finetune``) -- an indexed batch cannot be read without them.  Same
    numbers as the same batches gathered into contiguous tensors (whose statistics the step computes itself)."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd import token_store as TS
    from efficient_probing_amd.util.lars import LARS
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    g = torch.Generator().manual_seed(11)
    n_img = 3 * B + 3
    tokens = torch.randn(n_img, N_TOK, D, generator=g).to(DEV)
    labels = torch.randint(0, C, (n_img,), generator=g).to(DEV)
    store = TS.ResidentTokenStore.from_tensors(tokens, labels, seed=2)
    gathered = [(t[i.long()].contiguous(), y) for t, i, y in store.batches(B, epoch=0)]
    assert len(gathered) == 3
    args = Namespace(accum_iter=1, amp="none", lr=0.2, min_lr=0.0, warmup_epochs=0, epochs=2, output_dir="", suffix="t", resume="")
    models = []
    for feed in ("store", "gathered"):
        model = fix_grid(make_model(name))
        opt = LARS(model.head.parameters(), lr=0.0, weight_decay=0.0)
        data = store.loader(B, epoch=0) if feed == "store" else gathered
        EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), data, opt, torch.device(DEV), 0, NativeScalerWithGradNormCount(), args=args)
        models.append(model)
    eng = EF.get_engine(models[0])
    if getattr(eng, "_store_kinds", None):
        assert eng._store is store and store.__dict__.get("_tables"), f"{name}: the store's tables were not used"
    for (k, a), (_, b) in zip(models[0].head.state_dict().items(), models[1].head.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=2e-5, atol=2e-6), f"{name}: {k} differs by {float((a.float() - b.float()).abs().max()):.3e}"
    ev_store = EF.evaluate(store.loader(B, epoch=1, shuffle=False, drop_last=False), models[0], torch.device(DEV))
    ev_gath = EF.evaluate([(t[i.long()].contiguous(), y) for t, i, y in store.batches(B, epoch=1, shuffle=False, drop_last=False)],
                          models[0], torch.device(DEV))
    assert abs(ev_store["loss"] - ev_gath["loss"]) <= 1e-5 * max(1.0, abs(ev_gath["loss"])) and ev_store["acc1"] == ev_gath["acc1"]


@pytest.mark.parametrize("name", ["ep", "cae", "simpool", "clip", "aim", "cbam"])
def test_bf16_resident_store_epoch_equals_gathered_batches(name):
    """The same with the tokens STORED as bf16 (the protocol's dump format for the large encoders): the store's tables are then
    computed from the bf16 tokens, exactly what a step computes for a gathered bf16 batch."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd import token_store as TS
    from efficient_probing_amd.util.lars import LARS
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    g = torch.Generator().manual_seed(13)
    n_img = 2 * B + 5
    tokens = torch.randn(n_img, N_TOK, D, generator=g).to(DEV).to(torch.bfloat16)
    labels = torch.randint(0, C, (n_img,), generator=g).to(DEV)
    store = TS.ResidentTokenStore.from_tensors(tokens, labels, seed=4)
    assert store.dtype == "bfloat16"
    gathered = [(t[i.long()].contiguous(), y) for t, i, y in store.batches(B, epoch=0)]
    args = Namespace(accum_iter=1, amp="none", lr=0.2, min_lr=0.0, warmup_epochs=0, epochs=2, output_dir="", suffix="t", resume="")
    models = []
    for feed in ("store", "gathered"):
        model = fix_grid(make_model(name))
        opt = LARS(model.head.parameters(), lr=0.0, weight_decay=0.0)
        data = store.loader(B, epoch=0) if feed == "store" else gathered
        EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), data, opt, torch.device(DEV), 0, NativeScalerWithGradNormCount(), args=args)
        models.append(model)
    for (k, a), (_, b) in zip(models[0].head.state_dict().items(), models[1].head.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=2e-5, atol=2e-6), f"{name}: {k} differs by {float((a.float() - b.float()).abs().max()):.3e}"
