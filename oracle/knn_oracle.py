"""CPU oracle for the weighted k-NN classifier -- TEST INFRASTRUCTURE ONLY.

numpy restatement of reference engine_finetune.py:224-266 (``knn_classifier``): similarity = test @ train^T, the k
largest per row (descending), votes exp(sim / T) summed per neighbour label, classes ranked by vote (stable
descending order, i.e. ties resolved towards the lower class index), top-1 / top-5 hit rates in percent.

PARITY PIN: tests/golden/knn_fixtures.json holds (top1, top5) computed by the REAL reference function on seeded
inputs (tests/golden/make_golden.py); tests/test_knn_cpu.py checks this restatement against them.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import numpy as np


def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    n = np.sqrt((x.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32))
    return (x / np.maximum(n, eps)).astype(np.float32)


def knn_search(train: np.ndarray, test: np.ndarray, k: int):
    sim = test.astype(np.float32) @ train.astype(np.float32).T                      # :243
    # k largest, sorted descending; ties -> lower train index first (stable sort on the negated values)
    order = np.argsort(-sim, axis=1, kind="stable")[:, :k]                          # :244
    return np.take_along_axis(sim, order, axis=1), order.astype(np.int32)


def knn_vote(sims: np.ndarray, idx: np.ndarray, train_labels: np.ndarray, test_labels, k: int, T: float,
             num_classes: int):
    M = sims.shape[0]
    neigh = train_labels[idx[:, :k]]                                                # :245-246
    w = np.exp(sims[:, :k].astype(np.float32) / np.float32(T)).astype(np.float32)   # :250
    probs = np.zeros((M, num_classes), np.float32)
    for i in range(k):                                                              # :251-257 (sum over the k axis)
        np.add.at(probs, (np.arange(M), neigh[:, i]), w[:, i])
    pred = np.argsort(-probs, axis=1, kind="stable")[:, :5]                         # :258
    if test_labels is None:
        return None, None, pred
    correct = pred == np.asarray(test_labels).reshape(-1, 1)                        # :261
    return 100.0 * correct[:, :1].sum() / M, 100.0 * correct[:, :5].sum() / M, pred   # :262-266


def knn_classifier(train, train_labels, test, test_labels, k, T, num_classes=1000):
    sims, idx = knn_search(train, test, k)
    t1, t5, _ = knn_vote(sims, idx, np.asarray(train_labels), test_labels, k, T, num_classes)
    return t1, t5
