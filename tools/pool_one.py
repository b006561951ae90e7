#!/usr/bin/env python3
"""The two EP token passes launched back to back (GPU box; run under rocprofv3).  usage: pool_one.py [iters] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
Nn = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B, D, Q = 1024, 768, 8
dev = "cuda:0"
xs = [torch.randn(B, Nn, D, device=dev) for _ in range(3)]
cls = torch.randn(Q, D, device=dev) * float(os.environ.get("CLS_STD", 0.5))   # query scale: how peaked the softmax is
dP = torch.randn(B, Q, D, device=dev)
for i in range(it):
    P, S, ML = F_.pool_forward(xs[i % 3], cls, D ** -0.5)
    ML[:, :, 2] = 0.1
    F_.pool_backward(xs[i % 3], S, ML, dP, D ** -0.5)
torch.cuda.synchronize()
