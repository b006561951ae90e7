"""GPU: batches drawn in place from a resident token store (image_index) give exactly what a gathered
batch gives, for every pooling kernel family; the streaming loader delivers the stored data."""
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(40, 37, 64, 4), (48, 50, 768, 8), (24, 33, 1024, 8), (24, 20, 384, 1)],
                         ids=["generic-D64", "valu-768", "mfma-1024", "valu-q1"])
def test_indexed_pool_matches_gathered(shape):
    from efficient_probing_amd import functional as F_, _native
    M, Nn, D, Q = shape
    g = torch.Generator().manual_seed(0)
    store = torch.randn(M, Nn, D, generator=g).to(DEV)
    cls = (torch.randn(Q, D, generator=g) * 0.3).to(DEV)
    idx = torch.randperm(M, generator=g)[:M // 2].to(torch.int32).to(DEV)
    scale = D ** -0.5
    lib = _native.load()
    for mode in (0, 1, 2, 3):
        lib.ep_debug_force_generic_pool(mode)
        try:
            Pi, Si, MLi = F_.pool_forward(store, cls, scale, image_index=idx)
            Pg, Sg, MLg = F_.pool_forward(store[idx.long()].contiguous(), cls, scale)
            assert torch.equal(Pi, Pg) and torch.equal(Si, Sg) and torch.equal(MLi, MLg), mode
            dP = torch.randn(Pi.shape, generator=g).to(DEV)
            MLi[:, :, 2] = 0.1
            di = F_.pool_backward(store, Si, MLi, dP, scale, image_index=idx)
            dg = F_.pool_backward(store[idx.long()].contiguous(), Si, MLi, dP, scale)
            assert torch.equal(di, dg), mode
        finally:
            lib.ep_debug_force_generic_pool(0)


def test_resident_store_training_equals_gathered_batches(tmp_path):
    from efficient_probing_amd import probe_heads, token_store as TS
    from efficient_probing_amd.engine import ProbeHeadEngine
    rng = np.random.default_rng(3)
    Nn, D, Q, C = 20, 256, 8, 12
    w = TS.TokenStoreWriter(str(tmp_path), Nn, D, shard_images=40)
    tok = rng.standard_normal((100, Nn, D), dtype=np.float32); lab = rng.integers(0, C, 100)
    w.add(tok, lab); w.close()
    store = TS.ResidentTokenStore(str(tmp_path), DEV, world=1, rank=0, seed=5)
    assert store.num_images == 100 and store.tokens.shape == (100, Nn, D)
    assert np.array_equal(store.tokens.cpu().numpy(), tok)

    def make():
        class Enc(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.head = torch.nn.Linear(D, C)
        torch.manual_seed(0)
        e = Enc()
        probe_heads.build_probe_head(e, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
        return ProbeHeadEngine(e.head.to(DEV).train(), optimizer="lars", lr=0.3)
    e1, e2 = make(), make()
    n = 0
    for tokens, idx, tgt in store.batches(32, epoch=0):
        e1.train_step(tokens, tgt, image_index=idx)
        e2.train_step(tokens[idx.long()].contiguous(), tgt)
        n += 1
    assert n == 3
    for a, b in zip(e1.params_list, e2.params_list):
        assert torch.equal(a, b)
    assert e1.read_stats() == e2.read_stats()
    # two ranks see disjoint shards that together cover the store
    s0 = TS.ResidentTokenStore(str(tmp_path), DEV, world=2, rank=0)
    s1 = TS.ResidentTokenStore(str(tmp_path), DEV, world=2, rank=1)
    assert s0.num_images + s1.num_images == 100 and s0.num_images == 60
    # ... and run the same number of steps per epoch although they own 60 / 40 images
    assert len(list(s0.batches(16))) == len(list(s1.batches(16))) == 2 == TS.steps_per_epoch(TS.load_meta(str(tmp_path)), 2, 16)
    l0 = TS.StreamingTokenLoader(str(tmp_path), DEV, batch_size=16, world=2, rank=0)
    l1 = TS.StreamingTokenLoader(str(tmp_path), DEV, batch_size=16, world=2, rank=1)
    assert len(l0) == len(l1) == 2 and sum(1 for _ in l0) == sum(1 for _ in l1) == 2


def test_streaming_loader_delivers_the_store(tmp_path):
    from efficient_probing_amd import token_store as TS
    rng = np.random.default_rng(4)
    w = TS.TokenStoreWriter(str(tmp_path), 6, 64, shard_images=25)
    tok = rng.standard_normal((70, 6, 64), dtype=np.float32); lab = rng.integers(0, 5, 70)
    w.add(tok, lab); w.close()
    ld = TS.StreamingTokenLoader(str(tmp_path), DEV, batch_size=8)
    seen_t, seen_l = [], []
    for x, t in ld:
        seen_t.append(x.cpu().numpy().copy()); seen_l.append(t.cpu().numpy())
    assert len(seen_t) == len(ld) == 3 + 3 + 2          # per shard: 25//8, 25//8, 20//8
    want = np.concatenate([tok[0:24], tok[25:49], tok[50:66]])
    assert np.array_equal(np.concatenate(seen_t), want)
    assert np.array_equal(np.concatenate(seen_l), np.concatenate([lab[0:24], lab[25:49], lab[50:66]]))


def test_extract_features_from_a_dumped_store_feeds_knn(tmp_path):
    """dump (stub encoder) -> resident store -> extract_features (reference engine_finetune.py:168-222 signature): the
    features are the token means, equal to knn.mean_tokens of the stored tokens and to what the same encoder gives through
    ``token_fn`` on the images; knn_classifier runs on them."""
    from efficient_probing_amd import dump, engine_finetune as EF, knn, token_store as TS
    torch.manual_seed(0)
    proj = torch.nn.Conv2d(3, 64, kernel_size=4, stride=4).to(DEV).eval()
    token_fn = lambda x: proj(x).flatten(2).transpose(1, 2)                       # (B, 16, 64)
    g = torch.Generator().manual_seed(1)
    imgs = torch.randn(50, 3, 16, 16, generator=g)
    labs = torch.randint(0, 5, (50,), generator=g)
    loader = [(imgs[i:i + 16], labs[i:i + 16]) for i in range(0, 50, 16)]
    meta = dump.dump_tokens(loader, token_fn, str(tmp_path), dtype="float32", shard_images=32, device=torch.device(DEV))
    assert meta["total_images"] == 50
    store = TS.ResidentTokenStore(str(tmp_path), device=torch.device(DEV))
    st = EF.extract_features(store.loader(16, shuffle=False, drop_last=False), None, torch.device(DEV), return_targets_and_preds=True)
    assert st["features"].shape == (50, 64) and torch.equal(st["targets"], labs)
    want = knn.mean_tokens(store.tokens).cpu()
    assert torch.equal(st["features"], want)
    np.testing.assert_allclose(want.numpy(), store.tokens.float().mean(1).cpu().numpy(), rtol=1e-5, atol=1e-6)
    # the same features straight from the images through token_fn (3-tuple batches: batch[0] / batch[-1])
    st2 = EF.extract_features([(a, torch.zeros(len(a)), b) for a, b in loader], None, torch.device(DEV), return_targets_and_preds=True,
                              token_fn=token_fn)
    np.testing.assert_allclose(st2["features"].numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    top1, top5 = knn.knn_classifier(st["features"].to(DEV), st["targets"].to(DEV), st2["features"].to(DEV), st2["targets"].to(DEV),
                                    k=5, T=0.07, num_classes=5)
    assert top1 == 100.0                                                          # every image finds itself


def test_one_image_index_batch_into_a_strided_store():
    """the fast one-call step took the batch stride from the INDEX count -- a one-element index batch into a store
    whose image stride is not N * D (the `tokens[:, 1:]` view) then stepped by N * D and trained on the wrong image."""
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine
    Nn, D, Q, C, M = 20, 256, 8, 12, 6
    g = torch.Generator().manual_seed(11)
    full = torch.randn(M, Nn + 1, D, generator=g).to(DEV)
    store = full[:, 1:]                                      # image stride (Nn + 1) * D
    tgt = torch.tensor([5], dtype=torch.int64, device=DEV)

    def make():
        class Enc(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.head = torch.nn.Linear(D, C)
        torch.manual_seed(0)
        e = Enc()
        probe_heads.build_probe_head(e, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
        return ProbeHeadEngine(e.head.to(DEV).train(), optimizer="sgd", lr=0.3)
    e1, e2 = make(), make()
    for i in (4, 2):
        idx = torch.tensor([i], dtype=torch.int32, device=DEV)
        e1.train_step(store, tgt, image_index=idx)
        e2.train_step(store[i:i + 1].contiguous(), tgt)
    for a, b in zip(e1.params_list, e2.params_list):
        assert torch.equal(a, b)
    assert e1.read_stats() == e2.read_stats()
    # a batch of one leaves no gradient behind the BatchNorm (z = 0): what tells the images apart is the running mean
    assert torch.equal(e1.bn.running_mean, e2.bn.running_mean) and float(e1.bn.running_mean.abs().max()) > 0
    wrong = make()
    wrong.train_step(store[0:1].contiguous(), tgt)
    wrong.train_step(store[0:1].contiguous(), tgt)
    assert not torch.equal(wrong.bn.running_mean, e1.bn.running_mean)
