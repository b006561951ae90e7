"""Weighted k-NN classifier on the GPU (ep_knn_topk / ep_knn_vote through the C ABI) against the CPU oracle and the
hit rates of the real reference function.  Needs an MI355X (pytest -m gpu).  Neighbour indices are compared exactly
(the similarity is exact fp32 on both sides up to summation order, so rows whose k-th and (k+1)-th neighbours are
closer than 1e-6 are excused)."""
import json
import os

import numpy as np
import pytest
import torch

from cases import KNN_CASES, make_knn_inputs
from oracle import knn_oracle as KO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "knn_fixtures.json")))


def dev(a):
    return torch.from_numpy(a).to(DEV)


@pytest.mark.parametrize("name", sorted(KNN_CASES))
def test_search_and_vote_vs_oracle_and_reference(name):
    from efficient_probing_amd import knn
    inp = make_knn_inputs(name)
    kmax = max(r["k"] for r in FX[name])
    sims, idx = knn.knn_search(dev(inp["train"]), dev(inp["test"]), kmax)
    s, i = sims.cpu().numpy(), idx.cpu().numpy()
    so, io = KO.knn_search(inp["train"], inp["test"], kmax + 1)
    assert (np.diff(s, axis=1) <= 0).all()
    np.testing.assert_allclose(s, so[:, :kmax], rtol=0, atol=2e-6)
    # index parity wherever the ranking is not decided by sub-rounding differences
    gaps = np.abs(np.diff(so, axis=1))
    decided = np.concatenate([np.ones((len(s), 1), bool), gaps[:, :kmax - 1] > 4e-6], axis=1) & \
        np.concatenate([gaps[:, :kmax] > 4e-6], axis=1)
    assert (i[decided] == io[:, :kmax][decided]).all()
    assert decided.mean() > 0.95 or name == "dups"
    # the similarity each reported index carries is the true one
    true = np.einsum("md,mkd->mk", inp["test"], inp["train"][i])
    np.testing.assert_allclose(s, true, rtol=0, atol=2e-6)
    tol = 100.0 / len(inp["test_labels"]) + 1e-9
    for row in FX[name]:
        t1, t5, pred = knn.knn_vote(sims, idx, dev(inp["train_labels"]), dev(inp["test_labels"]), row["k"], row["T"], inp["C"])
        assert abs(t1 - row["top1"]) <= tol and abs(t5 - row["top5"]) <= tol, (row, t1, t5)
        o1, o5, op = KO.knn_vote(s, i, inp["train_labels"], inp["test_labels"], row["k"], row["T"], inp["C"])
        assert abs(t1 - o1) <= tol and abs(t5 - o5) <= tol
        assert (pred.cpu().numpy()[:, 0] == op[:, 0]).mean() > 0.99
    t1, t5 = knn.knn_classifier(dev(inp["train"]), dev(inp["train_labels"]), dev(inp["test"]), dev(inp["test_labels"]),
                                FX[name][1]["k"], FX[name][1]["T"], num_classes=inp["C"])
    assert abs(t1 - FX[name][1]["top1"]) <= tol and abs(t5 - FX[name][1]["top5"]) <= tol


def test_exact_ties_pick_the_lowest_indices_and_pathological_rows():
    from efficient_probing_amd import knn
    rng = np.random.default_rng(3)
    base = rng.standard_normal((50, 16)).astype(np.float32)
    train = np.concatenate([base, base, base], axis=0)              # every row three times: exact ties everywhere
    test = base[:8].copy()
    sims, idx = knn.knn_search(dev(train), dev(test), 4)
    i = idx.cpu().numpy()
    for r in range(8):
        assert list(i[r, :3]) == [r, r + 50, r + 100]               # the tied self-matches, lowest index first
    # a row with thousands of identical similarities (all-equal train features): still k distinct, lowest indices
    train = np.ones((5000, 8), np.float32)
    sims, idx = knn.knn_search(dev(train), dev(np.ones((3, 8), np.float32)), 10)
    assert (idx.cpu().numpy() == np.arange(10)[None, :]).all() and (sims.cpu().numpy() == 8.0).all()


def test_sweep_normalize_and_token_mean():
    from efficient_probing_amd import knn
    inp = make_knn_inputs("small")
    raw_tr, raw_te = inp["train"] * 3.0, inp["test"] * 0.5                                 # un-normalised copies
    res = knn.knn_sweep(dev(raw_tr), dev(inp["train_labels"]), dev(raw_te), dev(inp["test_labels"]),
                        ks=(5, 20, 200), temperatures=(0.07,), num_classes=inp["C"])
    tol = 100.0 / len(inp["test_labels"]) + 1e-9
    for row in FX["small"]:
        if (row["T"], row["k"]) in res:
            assert abs(res[(row["T"], row["k"])][0] - row["top1"]) <= tol
    n = knn.l2_normalize(dev(raw_tr)).cpu().numpy()
    np.testing.assert_allclose(n, KO.l2_normalize(raw_tr), rtol=2e-6, atol=1e-7)
    x = torch.randn(6, 50, 128, device=DEV)
    np.testing.assert_allclose(knn.mean_tokens(x).cpu().numpy(), x.mean(dim=1).cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_full_size_search_is_fast_and_consistent():
    """ImageNet-sized gallery slice: 200k x 768 gallery, 2048 queries, k = 200: sorted output, exact self-retrieval."""
    from efficient_probing_amd import knn
    g = torch.Generator(device=DEV).manual_seed(0)
    train = knn.l2_normalize(torch.randn(200_000, 768, device=DEV, generator=g))
    sel = torch.randperm(200_000, device=DEV, generator=g)[:2048]
    test = train[sel].contiguous()
    sims, idx = knn.knn_search(train, test, 200)
    assert (idx[:, 0].long() == sel).all()                          # each query finds itself first (sim = 1)
    assert (sims[:, :-1] >= sims[:, 1:]).all()
    ref = (test[:64] @ train.t()).topk(200, dim=1).values            # stock fp32 GEMM + topk as an independent check
    np.testing.assert_allclose(sims[:64].cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=3e-6)


def test_fast_select_path_ties_fallback_and_ordered_rows():
    """Rows long enough for the two-pass selection (group maxima -> threshold -> gather; csrc/ep_knn.hip): the same exact
    result and tie rule (value descending, lowest index first) as the radix passes it replaces, which still take over when
    the candidates do not come together."""
    from efficient_probing_amd import knn
    rng = np.random.default_rng(11)
    # (1) every gallery row 16 times: exact ties everywhere, also across the k-th place
    base = rng.standard_normal((4096, 16)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)              # unit rows: a row's best match is itself
    train = np.tile(base, (16, 1))                                   # row j + 4096 c == row j
    test = base[:32].copy()
    k = 50
    sims, idx = knn.knn_search(dev(train), dev(test), k)
    s, i = sims.cpu().numpy(), idx.cpu().numpy()
    for r in range(32):
        assert list(i[r, :16]) == [r + 4096 * c for c in range(16)]
        order = np.lexsort((i[r], -s[r]))
        assert (order == np.arange(k)).all()                         # sorted by (value descending, index ascending)
        # tie groups: complete ones hold all 16 copies, the cut one the LOWEST copies
        for v in np.unique(s[r]):
            grp = i[r][s[r] == v]
            j = grp[0] % 4096
            assert (grp == j + 4096 * np.arange(len(grp))).all()
    true = np.einsum("md,mkd->mk", test, train[i])
    np.testing.assert_allclose(s, true, rtol=0, atol=2e-5)
    ref = (dev(test) @ dev(train).t()).topk(k, dim=1).values.cpu().numpy()
    np.testing.assert_allclose(s, ref, rtol=0, atol=2e-5)
    # (2) all-equal similarities: no element above any threshold -> the radix passes, lowest indices
    ones = np.ones((40000, 8), np.float32)
    sims, idx = knn.knn_search(dev(ones), dev(np.ones((3, 8), np.float32)), 10)
    assert (idx.cpu().numpy() == np.arange(10)[None, :]).all() and (sims.cpu().numpy() == 8.0).all()
    # (3) similarities increasing along the row: the top k sit at its very end
    ramp = np.linspace(-1.0, 1.0, 70001, dtype=np.float32)[:, None] * np.ones((1, 4), np.float32)
    sims, idx = knn.knn_search(dev(ramp), dev(np.ones((2, 4), np.float32)), 200)
    assert (idx.cpu().numpy() == np.arange(70000, 70000 - 200, -1)[None, :]).all()
    # (4) k at the capacity of the neighbour list
    g = torch.Generator(device=DEV).manual_seed(2)
    tr = knn.l2_normalize(torch.randn(100_000, 64, device=DEV, generator=g))
    te = knn.l2_normalize(torch.randn(16, 64, device=DEV, generator=g))
    sims, idx = knn.knn_search(tr, te, 1024)
    ref = (te @ tr.t()).topk(1024, dim=1)
    np.testing.assert_allclose(sims.cpu().numpy(), ref.values.cpu().numpy(), rtol=0, atol=2e-6)
    assert (idx.long() == ref.indices).float().mean() > 0.98        # (near-ties may swap between the two contractions)
