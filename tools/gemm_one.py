#!/usr/bin/env python3
"""One contraction of the head tail, launched back to back (GPU box): run under rocprofv3 --kernel-trace --stats to read
its device duration.  usage: gemm_one.py {y|logits|dz|dP|dWc|dWv} [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
op = sys.argv[1]; it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, D, Q, C = int(os.environ.get("B", 1024)), int(os.environ.get("D", 768)), 8, 1000
dev = "cuda:0"
P = torch.randn(B, Q, D, device=dev); Wv = torch.randn(D, D, device=dev) * 0.03
z = torch.randn(B, D, device=dev); Wc = torch.randn(C, D, device=dev) * 0.03; bc = torch.randn(C, device=dev)
dl = torch.randn(B, C, device=dev); dy = torch.randn(B, D, device=dev)
dWc = torch.empty_like(Wc); dbc = torch.empty(C, device=dev); dWv = torch.empty_like(Wv)
big = torch.empty(64 << 20, device=dev)          # 256 MiB: flush L2 / Infinity Cache between launches
fn = {"y": lambda: F_.project_forward(P, Wv), "logits": lambda: F_.linear_forward(z, Wc, bc),
      "dz": lambda: F_.linear_backward(dl, z, Wc, True, None, None, False),
      "dP": lambda: F_.project_backward(dy, None, P, Wv, None, True, None, False, False),
      "dWc": lambda: F_.linear_backward(dl, z, Wc, False, dWc, None, False),
      "dWv": lambda: F_.project_backward(dy, None, P, Wv, None, False, dWv, False, True)}[op]
for i in range(it):
    if os.environ.get("FLUSH"): big.zero_()
    fn()
torch.cuda.synchronize()
