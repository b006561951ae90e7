#!/usr/bin/env python3
"""Large-M throughput of the exact-fp32 matrix-core contraction (GPU box): the AbMILP-sized GEMMs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
dev = "cuda:0"
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / it
for (M, K, N) in [(65536, 1152, 3456), (65536, 1152, 1152), (65536, 768, 2304), (16384, 768, 768)]:
    z = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    t = timeit(lambda: F_.linear_forward(z, W, b))
    t2 = timeit(lambda: torch.addmm(b, z, W.t()))
    print(json.dumps({"M": M, "K": K, "N": N, "ms": round(t * 1e3, 3), "TFLOPs": round(2 * M * K * N / t / 1e12, 1),
                      "torch_ms": round(t2 * 1e3, 3), "torch_TFLOPs": round(2 * M * K * N / t2 / 1e12, 1),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("EP_")}}))
