#!/usr/bin/env python3
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
dev = "cuda:0"
def timeit(fn, it=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / it, 1)
for B, K, C in [(1024, 96, 1000), (1024, 192, 1000), (1024, 384, 1000), (1024, 768, 1000), (1024, 1536, 1000), (1024, 3072, 1000), (2048, 768, 1000), (4096, 768, 1000), (512, 768, 1000), (256, 768, 1000)]:
    z = torch.randn(B, K, device=dev); Wc = torch.randn(C, K, device=dev) * 0.03; bc = torch.randn(C, device=dev)
    t = timeit(lambda: F_.linear_forward(z, Wc, bc))
    print(f"B={B} K={K} C={C}: {t} us  -> {2*B*K*C/t/1e6:.1f} TFLOP/s")
