#!/usr/bin/env python3
"""ImageNet-sized weighted k-NN evaluation on the GPU box: 1.28 M x D gallery, Q queries, k = 200, full k/T sweep."""
import argparse, os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import knn
ap = argparse.ArgumentParser()
ap.add_argument("--gallery", type=int, default=1_281_167); ap.add_argument("--queries", type=int, default=10_000)
ap.add_argument("--dim", type=int, default=768); ap.add_argument("--k", type=int, default=200)
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
train = knn.l2_normalize(torch.randn(a.gallery, a.dim, device=dev, generator=g))
test = knn.l2_normalize(torch.randn(a.queries, a.dim, device=dev, generator=g))
ltr = torch.randint(0, 1000, (a.gallery,), device=dev, generator=g); lte = torch.randint(0, 1000, (a.queries,), device=dev, generator=g)
knn.knn_search(train, test[:256], a.k); torch.cuda.synchronize()
t0 = time.perf_counter(); sims, idx = knn.knn_search(train, test, a.k); torch.cuda.synchronize(); t_search = time.perf_counter() - t0
t0 = time.perf_counter()
res = {(T, k): knn.knn_vote(sims, idx, ltr, lte, k, T)[:2] for T in (0.02, 0.07, 0.2) for k in (5, 10, 15, 20, 50, 100, 200)}
torch.cuda.synchronize(); t_vote = time.perf_counter() - t0
# GEMM alone for the same shape (one chunk)
from efficient_probing_amd import functional as F_
rows = 400
torch.cuda.synchronize(); t0 = time.perf_counter(); F_.linear_forward(test[:rows], train, None); torch.cuda.synchronize(); t_g = time.perf_counter() - t0
print(json.dumps({"gallery": a.gallery, "queries": a.queries, "dim": a.dim, "k": a.k, "search_s": round(t_search, 3),
                  "queries_per_s": round(a.queries / t_search, 1), "vote_sweep_21_settings_s": round(t_vote, 3),
                  "gemm_TFLOPs_400rows": round(2 * rows * a.gallery * a.dim / t_g / 1e12, 1),
                  "search_TFLOPs_equiv": round(2 * a.queries * a.gallery * a.dim / t_search / 1e12, 1)}))
