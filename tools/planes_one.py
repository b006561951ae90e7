#!/usr/bin/env python3
"""One planes contraction launched back to back (GPU box; run under rocprofv3).  usage: planes_one.py {logits|dz|split} [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
op = sys.argv[1]; it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, D, Q, C = 1024, 768, 8, 1000
dev = "cuda:0"
z = torch.randn(B, D, device=dev); Wc = torch.randn(C, D, device=dev) * 0.03; bc = torch.randn(C, device=dev)
dl = torch.randn(B, C, device=dev)
pn, pt = F_.planes_split(Wc, True, True)
if op == "logits": fn = lambda: F_.matmul_planes(z, pn, C, bias=bc)
elif op == "dz": fn = lambda: F_.matmul_planes(dl, pt, D)
elif op == "split": fn = lambda: F_.planes_split(Wc, True, True)
for i in range(it): fn()
torch.cuda.synchronize()
