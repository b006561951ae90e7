#!/usr/bin/env python3
"""One planes contraction launched back to back (GPU box; run under rocprofv3).  usage: planes_one.py {logits|dz|split|mm} [iters] [M K N]  (mm: an M x K activation against an N x K weight, and the f32 kernel on the same)"""
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
if op == "mm":
    M, K, Nw = (int(v) for v in sys.argv[3:6])
    A = torch.randn(M, K, device=dev); W = torch.randn(Nw, K, device=dev) * 0.03
    pw, _ = F_.planes_split(W)
    got = F_.matmul_planes(A, pw, Nw); ref = A.double() @ W.double().t()
    f32 = F_.linear_forward(A, W, None)
    print(f"mm {M}x{K}x{Nw}: planes err {float((got.double() - ref).abs().max() / ref.abs().max()):.2e}, f32 kernel err {float((f32.double() - ref).abs().max() / ref.abs().max()):.2e}")
    def fn():
        F_.matmul_planes(A, pw, Nw); F_.linear_forward(A, W, None)
for i in range(it): fn()
torch.cuda.synchronize()
