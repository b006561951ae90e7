"""CPU: the token dump entry (efficient_probing_amd.dump -- counterpart of reference tools/dump_tokens.py:60-98) drives a stub
encoder + loader into the sharded store, and what comes back out of the store is what went in."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubEncoder(torch.nn.Module):
    """A frozen "backbone": 4x4 patches of an (B, 3, 16, 16) image, linearly embedded -> (B, 16, 24) tokens."""
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.proj = torch.nn.Conv2d(3, 24, kernel_size=4, stride=4)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def make_loader(n=37, batch=8):
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn(n, 3, 16, 16, generator=g)
    labs = torch.randint(0, 10, (n,), generator=g)
    extra = torch.arange(n)
    # three-element batches: read as batch[0] / batch[-1] (reference engine_finetune.py:185-186)
    return imgs, labs, [(imgs[i:i + batch], extra[i:i + batch], labs[i:i + batch]) for i in range(0, n, batch)]


@pytest.mark.parametrize("dtype", ["float32", "bf16"])
def test_dump_roundtrip_with_stub_encoder(tmp_path, dtype):
    from efficient_probing_amd import dump, token_store as TS
    enc = StubEncoder().eval()
    imgs, labs, loader = make_loader()
    meta = dump.dump_tokens(loader, enc, str(tmp_path), dtype=dtype, shard_images=16)
    assert meta["total_images"] == 37 and meta["num_tokens"] == 16 and meta["dim"] == 24
    assert [s["images"] for s in meta["shards"]] == [16, 16, 5]
    assert meta["dtype"] == ("float32" if dtype == "float32" else "bfloat16")
    with torch.no_grad():
        want = enc(imgs)
    got_t, got_l = [], []
    for s in TS.load_meta(str(tmp_path))["shards"]:
        tok, lab = TS.open_shard(str(tmp_path), meta, s)
        got_t.append(TS._to_torch(np.asarray(tok), meta["dtype"]).float()); got_l.append(torch.from_numpy(lab))
    got_t, got_l = torch.cat(got_t), torch.cat(got_l)
    assert torch.equal(got_l, labs)
    if dtype == "float32":
        assert torch.equal(got_t, want)
    else:
        assert torch.equal(got_t, want.to(torch.bfloat16).float())          # stored rounded to nearest even, nothing else
    # max_images cuts inside a batch
    meta2 = dump.dump_tokens(loader, enc, str(tmp_path / "cut"), dtype=dtype, shard_images=16, max_images=11)
    assert meta2["total_images"] == 11
    with pytest.raises(ValueError):
        dump.dump_tokens([], enc, str(tmp_path / "none"))


def test_feature_maps_are_flattened_like_the_reference():
    """tools/dump_tokens.py:51-53: (B, H, W, C) and (B, C, H, W) encoder outputs become (B, H*W, C) token rows."""
    from efficient_probing_amd.dump import token_view
    f = torch.arange(2 * 4 * 4 * 3, dtype=torch.float32).reshape(2, 4, 4, 3)     # channels last
    assert torch.equal(token_view(f), f.reshape(2, 16, 3))
    g = f.permute(0, 3, 1, 2).contiguous()                                       # channels first: (B, C, H, W) with C < H == W ... a > d is false
    assert token_view(torch.zeros(2, 8, 4, 4)).shape == (2, 16, 8)               # (B, C, H, W), C > W
    with pytest.raises(ValueError):
        token_view(torch.zeros(3, 4))


def test_cli_synthetic_and_reference_npz(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = tmp_path / "syn"
    r = subprocess.run([sys.executable, "-m", "efficient_probing_amd.dump", "--out", str(out), "--synthetic", "20", "--tokens", "6",
                        "--dim", "16", "--dtype", "bf16", "--shard-images", "8", "--batch", "7"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    meta = json.load(open(out / "meta.json"))
    assert meta["total_images"] == 20 and meta["dtype"] == "bfloat16" and [s["images"] for s in meta["shards"]] == [8, 8, 4]
    # a dump in the reference's own .npz layout (keys tokens / images / names)
    tok = np.random.default_rng(0).standard_normal((5, 6, 16)).astype(np.float32)
    np.savez_compressed(tmp_path / "ref.npz", tokens=tok, images=np.zeros((5, 3, 2, 2), np.float32), names=np.array(list("abcde")))
    out2 = tmp_path / "npz"
    r = subprocess.run([sys.executable, "-m", "efficient_probing_amd.dump", "--out", str(out2), "--from-npz", str(tmp_path / "ref.npz"),
                        "--dtype", "f32"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    from efficient_probing_amd import token_store as TS
    m2 = TS.load_meta(str(out2))
    t2, _ = TS.open_shard(str(out2), m2, m2["shards"][0])
    assert np.array_equal(np.asarray(t2), tok)
