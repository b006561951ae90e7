"""Training-free weighted k-NN evaluation on frozen features, native on MI355X.

Call surface of the reference (engine_finetune.py:224-266, main_linprobe.py:411-465):

    knn_classifier(train_features, train_labels, test_features, test_labels, k, T,
                   use_cuda=True, num_classes=1000, num_chunks=500) -> (top1 %, top5 %)

plus the two pieces it is made of, so that the reference's sweep over k in {5,...,200} and over temperatures
searches ONCE (``knn_search`` for the largest k) and votes many times (``knn_vote`` on prefixes):

    sims, idx = knn_search(train_features, test_features, kmax)
    top1, top5, pred = knn_vote(sims, idx, train_labels, test_labels, k, T, num_classes)

Similarities are exact fp32 (f32 matrix cores), the selection is an exact radix select; ``num_chunks`` is accepted
for signature compatibility only (the result of the reference does not depend on it).  No CPU path.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _native as N
from . import functional as F_


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """torch.nn.functional.normalize(x, dim=1, p=2) (reference main_linprobe.py:441-442)."""
    x = F_._f32c(x, "features")
    out = torch.empty_like(x)
    N.check(N.load().ep_l2_normalize(x.data_ptr(), x.shape[0], x.shape[1], eps, out.data_ptr(),
                                     N.current_stream_ptr(x.device)), "ep_l2_normalize")
    return out


def mean_tokens(x: torch.Tensor, image_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, N, D) tokens -> (B, D) mean over tokens (reference extract_features, engine_finetune.py:205-206), computed
    by the streaming token pass: a zero query gives uniform attention, i.e. the mean.  With ``image_index`` (int32 (B,)),
    ``x`` is a resident token store and the batch is read from it in place."""
    zero = torch.zeros((1, x.shape[-1]), device=x.device, dtype=torch.float32)
    P, _, _ = F_.pool_forward(x, zero, 1.0, image_index=image_index)
    return P[:, 0, :]


def knn_search(train_features: torch.Tensor, test_features: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(sims (M, k) descending, idx (M, k) int32) of the k most similar train rows of every test row."""
    lib = N.load()
    tr = F_._f32c(train_features, "train_features")
    te = F_._f32c(test_features, "test_features")
    if tr.dim() != 2 or te.dim() != 2 or tr.shape[1] != te.shape[1]:
        raise ValueError(f"features must be (n, D) with equal D, got {tuple(tr.shape)} and {tuple(te.shape)}")
    M, D = te.shape
    n_train = tr.shape[0]
    nbytes = lib.ep_knn_workspace_bytes_ex(M, n_train, D)
    ws = torch.empty(nbytes, device=te.device, dtype=torch.uint8)
    sims = torch.empty((M, k), device=te.device, dtype=torch.float32)
    idx = torch.empty((M, k), device=te.device, dtype=torch.int32)
    N.check(lib.ep_knn_topk(te.data_ptr(), tr.data_ptr(), M, n_train, D, k, sims.data_ptr(), idx.data_ptr(), k,
                            ws.data_ptr(), nbytes, N.current_stream_ptr(te.device)), "ep_knn_topk")
    return sims, idx


def knn_vote(sims: torch.Tensor, idx: torch.Tensor, train_labels: torch.Tensor, test_labels: Optional[torch.Tensor],
             k: int, T: float, num_classes: int = 1000):
    """(top1 %, top5 %, pred (M, 5) int32) from the first k columns of a neighbour list."""
    lib = N.load()
    M, ld = sims.shape
    if k > ld:
        raise ValueError(f"k={k} exceeds the searched neighbour count {ld}")
    labels = train_labels.to(device=sims.device, dtype=torch.int64).contiguous()
    tgt = test_labels.to(device=sims.device, dtype=torch.int64).contiguous() if test_labels is not None else None
    pred = torch.zeros((M, 5), device=sims.device, dtype=torch.int32)
    counts = torch.zeros(2, device=sims.device, dtype=torch.float32)
    N.check(lib.ep_knn_vote(sims.data_ptr(), idx.data_ptr(), ld, labels.data_ptr(), M, k, float(T), num_classes,
                            F_._ptr(tgt), pred.data_ptr(), counts.data_ptr(), N.current_stream_ptr(sims.device)),
            "ep_knn_vote")
    c = counts.tolist()
    return c[0] * 100.0 / M, c[1] * 100.0 / M, pred


@torch.no_grad()
def knn_classifier(train_features, train_labels, test_features, test_labels, k, T, use_cuda=True, num_classes=1000,
                   num_chunks=500):
    if not use_cuda:
        raise RuntimeError("knn_classifier (native): there is no CPU path")
    sims, idx = knn_search(train_features, test_features, k)
    top1, top5, _ = knn_vote(sims, idx, train_labels, test_labels, k, T, num_classes)
    return top1, top5


@torch.no_grad()
def knn_sweep(train_features, train_labels, test_features, test_labels, ks: Sequence[int] = (5, 10, 15, 20, 50, 100, 200),
              temperatures: Sequence[float] = (0.07,), num_classes: int = 1000, normalize: bool = True):
    """The reference's evaluation grid (main_linprobe.py:458-465) from ONE neighbour search: {(T, k): (top1, top5)}."""
    if normalize:
        train_features, test_features = l2_normalize(train_features), l2_normalize(test_features)
    sims, idx = knn_search(train_features, test_features, max(ks))
    return {(T, k): knn_vote(sims, idx, train_labels, test_labels, k, T, num_classes)[:2]
            for T in temperatures for k in ks}
