"""CPU tests of the two 'next' rows that are pure host logic: the token-store format and the
checkpoint / export dictionaries (reference util/misc.py:304-393, tools/export_ep_heads.py:124-138)."""
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch


class Enc(torch.nn.Module):
    def __init__(self, dim=64, classes=10):
        super().__init__()
        self.head = torch.nn.Linear(dim, classes)


def make_head(q=4, d_out=1):
    from efficient_probing_amd import probe_heads
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=q, d_out=d_out, nb_classes=10))
    return enc


def test_token_store_roundtrip_and_rank_partition(tmp_path):
    from efficient_probing_amd import token_store as TS
    rng = np.random.default_rng(0)
    w = TS.TokenStoreWriter(str(tmp_path), num_tokens=5, dim=8, shard_images=7)
    all_t, all_l = [], []
    for n in (4, 9, 6):                                   # ragged adds crossing shard boundaries
        t = rng.standard_normal((n, 5, 8), dtype=np.float32); l = rng.integers(0, 10, n)
        w.add(t, l); all_t.append(t); all_l.append(l)
    meta = w.close()
    all_t, all_l = np.concatenate(all_t), np.concatenate(all_l)
    assert meta["total_images"] == 19 and [s["images"] for s in meta["shards"]] == [7, 7, 5]
    meta2 = TS.load_meta(str(tmp_path))
    got_t, got_l = [], []
    for s in meta2["shards"]:
        tok, lab = TS.open_shard(str(tmp_path), meta2, s)
        got_t.append(np.asarray(tok)); got_l.append(lab)
    assert np.array_equal(np.concatenate(got_t), all_t) and np.array_equal(np.concatenate(got_l), all_l)
    r0 = TS.shards_of_rank(meta2, 2, 0); r1 = TS.shards_of_rank(meta2, 2, 1)
    assert [s["index"] for s in r0] == [0, 2] and [s["index"] for s in r1] == [1]
    # empty store and bad directory
    w2 = TS.TokenStoreWriter(str(tmp_path / "empty"), 5, 8)
    assert w2.close()["total_images"] == 0
    (tmp_path / "bad").mkdir()
    (tmp_path / "bad" / "meta.json").write_text("{}")
    with pytest.raises(ValueError):
        TS.load_meta(str(tmp_path / "bad"))


def test_uneven_shards_give_every_rank_the_same_number_of_steps(tmp_path):
    """Whole shards are dealt round-robin, so two ranks can own 60 / 40 images; each engine step ends in a blocking
    all-reduce, so both must run the same number of steps (the reference: DistributedSampler, main_linprobe.py:286-287)."""
    from efficient_probing_amd import token_store as TS
    rng = np.random.default_rng(2)
    w = TS.TokenStoreWriter(str(tmp_path), num_tokens=3, dim=8, shard_images=40)
    w.add(rng.standard_normal((100, 3, 8), dtype=np.float32), rng.integers(0, 10, 100))     # shards of 40, 40, 20
    meta = w.close()
    own = [sum(s["images"] for s in TS.shards_of_rank(meta, 2, r)) for r in (0, 1)]
    assert own == [60, 40]
    assert TS.steps_per_epoch(meta, 2, 16) == 2                       # min(60 // 16, 40 // 16)
    assert TS.steps_per_epoch(meta, 2, 16, per_shard=True) == 2       # rank 0: 40//16 + 20//16 = 3, rank 1: 2
    assert TS.steps_per_epoch(meta, 1, 16) == 6
    assert TS.steps_per_epoch(meta, 3, 16) == 1                       # 40, 40, 20 images
    for B in (7, 16, 20, 33):
        counts = []
        for r in (0, 1):
            st = TS.ResidentTokenStore(str(tmp_path), "cpu", world=2, rank=r)
            got = list(st.batches(B, epoch=1))
            counts.append(len(got))
            assert len(got) == st.num_batches(B)
            for tokens, idx, tgt in got:
                assert idx.numel() == B and int(idx.max()) < st.num_images
        assert counts[0] == counts[1] == min(60 // B, 40 // B)
    # one rank: the whole store, drop_last as before
    st = TS.ResidentTokenStore(str(tmp_path), "cpu")
    assert len(list(st.batches(16))) == 6 and st.num_batches(16, drop_last=False) == 7


def test_reads_reference_npz_dump(tmp_path):
    """tools/dump_tokens.py:95-98 writes tokens/images/names into one .npz."""
    from efficient_probing_amd import token_store as TS
    tok = np.random.default_rng(1).standard_normal((3, 4, 8)).astype(np.float32)
    np.savez(tmp_path / "dump.npz", tokens=tok, images=np.zeros((3, 2, 2, 3), np.uint8), names=np.array(["a", "b", "c"]))
    t, names = TS.read_reference_npz(str(tmp_path / "dump.npz"))
    assert np.array_equal(t, tok) and names == ["a", "b", "c"]


def test_checkpoint_format_and_resume_semantics(tmp_path):
    from efficient_probing_amd import checkpoint as CK
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    enc = make_head()
    opt = torch.optim.SGD(enc.head.parameters(), lr=0.1)
    args = Namespace(output_dir=str(tmp_path), suffix="run", resume="")
    scaler = NativeScalerWithGradNormCount()
    path = CK.save_model(args, 3, enc, enc.head, opt, scaler, {"test_acc1": 12.5})
    assert path.name == "checkpoint-run_3.pth"
    best = CK.save_model(args, 3, enc, enc.head, opt, scaler, {"test_acc1": 12.5}, filename_tag="best")
    assert best.name == "checkpoint-best.pth"
    ck = torch.load(path, weights_only=False)
    assert ck["saved_module"] == "head" and ck["epoch"] == 3
    assert sorted(ck["model"]) == ["0.cls_token", "0.v.weight", "1.num_batches_tracked", "1.running_mean",
                                   "1.running_var", "2.bias", "2.weight"]
    # resume into a fresh model: head-only checkpoint is routed into model.head, epoch counter advances
    enc2 = make_head()
    opt2 = torch.optim.SGD(enc2.head.parameters(), lr=0.1)
    args2 = Namespace(resume=str(path), start_epoch=0)
    stats = CK.load_model(args2, enc2, opt2, NativeScalerWithGradNormCount())
    assert stats == {"test_acc1": 12.5} and args2.start_epoch == 4
    for a, b in zip(enc.head.state_dict().values(), enc2.head.state_dict().values()):
        assert torch.equal(a, b)
    # a checkpoint that matches nothing must raise, not silently continue
    torch.save({"model": {"nothing.weight": torch.zeros(1)}, "epoch": 1}, tmp_path / "junk.pth")
    with pytest.raises(RuntimeError, match="matched 0"):
        CK.load_model(Namespace(resume=str(tmp_path / "junk.pth"), start_epoch=0), make_head(), None, None)
    # eval mode does not touch the optimizer / epoch
    args3 = Namespace(resume=str(path), start_epoch=0, eval=True)
    assert CK.load_model(args3, make_head(), opt2, None) is None and args3.start_epoch == 0


def test_export_format(tmp_path):
    from efficient_probing_amd import checkpoint as CK
    enc = make_head(q=4)
    meta = {"method": "dinov2", "arch": "vitb14", "cls_features": "ep", "ep_queries": 4, "d_out": 1, "head_epoch": 7}
    dst = CK.export_head(enc.head, meta, str(tmp_path), "dinov2-vitb14")
    assert sorted(os.listdir(dst)) == ["config.json", "ep_head.pth"]
    man = json.load(open(tmp_path / "manifest.json"))
    assert man["heads"][0]["file"] == "dinov2-vitb14/ep_head.pth"
    assert man["heads"][0]["params_incl_bn_stats"] == sum(v.numel() for v in enc.head.state_dict().values())
    enc2 = make_head(q=4)
    got = CK.load_exported_head(enc2.head, os.path.join(dst, "ep_head.pth"))
    assert got["ep_queries"] == 4
    assert torch.equal(enc2.head[0].cls_token, enc.head[0].cls_token)
    # the exported file is also accepted by load_model (reference tools read ck['state_dict'])
    enc3 = make_head(q=4)
    CK.load_model(Namespace(resume=os.path.join(dst, "ep_head.pth"), start_epoch=0), enc3, None, None)
    assert torch.equal(enc3.head[2].weight, enc.head[2].weight)


def test_bf16_token_store_roundtrip(tmp_path):
    """bf16 storage: values are the round-to-nearest-even bf16 of the fp32 tokens, shards hold raw 16-bit patterns."""
    import torch
    from efficient_probing_amd import token_store as TS
    rng = np.random.default_rng(3)
    w = TS.TokenStoreWriter(str(tmp_path), num_tokens=3, dim=16, shard_images=4, dtype="bfloat16")
    t = rng.standard_normal((6, 3, 16), dtype=np.float32)
    w.add(t, np.arange(6))
    meta = w.close()
    assert meta["dtype"] == "bfloat16" and [s["images"] for s in meta["shards"]] == [4, 2]
    tok, lab = TS.open_shard(str(tmp_path), meta, meta["shards"][0])
    assert tok.dtype == np.int16 and os.path.getsize(tmp_path / "tokens-00000.bin") == 4 * 3 * 16 * 2
    back = TS._to_torch(np.asarray(tok), "bfloat16").float().numpy()
    want = torch.from_numpy(t[:4]).to(torch.bfloat16).float().numpy()
    assert np.array_equal(back, want) and np.abs(back - t[:4]).max() < 2 ** -7 * np.abs(t).max()
    with pytest.raises(ValueError):
        TS.TokenStoreWriter(str(tmp_path / "x"), num_tokens=3, dim=12, dtype="bfloat16")
    with pytest.raises(NotImplementedError):
        TS.TokenStoreWriter(str(tmp_path / "y"), num_tokens=3, dim=16, dtype="float16")
