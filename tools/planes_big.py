#!/usr/bin/env python3
"""Planes contractions at the wide-row sizes, event-timed (GPU box).  usage: planes_big.py [M K N]...  (default: the c5 shapes)
Prints per shape: microseconds per launch, effective TFLOP/s (2 M N K), relative error against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
dev = "cuda:0"
args = [int(v) for v in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(1024, 4096, 4096), (1024, 512, 4096), (1024, 4096, 512), (1024, 1000, 4096), (1024, 4096, 1000), (1024, 768, 3072), (1024, 3072, 768)]
for M, K, N in shapes:
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03
    pw, _ = F_.planes_split(W)
    got = F_.matmul_planes(A, pw, N)
    ref = A.double() @ W.double().t()
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    for _ in range(5): F_.matmul_planes(A, pw, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it): F_.matmul_planes(A, pw, N)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"planes {M}x{K}x{N}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  err {err:.2e}  {os.environ.get('EP_PLANES_TILE', '')}", flush=True)
