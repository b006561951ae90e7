#!/usr/bin/env python3
"""The all-matrix-core forward pass (Q > 8) alone, event-timed (GPU box).  usage: mm_fwd_time.py [Q] [D] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_, _native as N_
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 12
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
Nn = int(sys.argv[3]) if len(sys.argv) > 3 else 256
B = 1024
dev = "cuda:0"
xs = [torch.randn(B, Nn, D, device=dev) for _ in range(4)]
cls = torch.randn(Q, D, device=dev) * 0.5
for i in range(8): F_.pool_forward(xs[i % 4], cls, D ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20): F_.pool_forward(xs[i % 4], cls, D ** -0.5)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
name = N_.load().ep_pool_kernel_name(B, Nn, D, Q, 0).decode()
print(f"{os.environ.get('EP_HIP_LIB', 'default').split('_')[-1]:>12s} {name} Q={Q} D={D} N={Nn}: {us:7.1f} us  {B * Nn * D * 4 / us / 1e6:5.2f} TB/s")
