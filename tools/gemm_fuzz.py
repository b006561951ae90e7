import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from efficient_probing_amd import functional as F_
torch.manual_seed(0)
dev = "cuda:0"
bad = 0
for (B, Dp, C) in [(64, 64, 64), (1024, 768, 1000), (37, 100, 52), (197*4, 128, 384), (5, 36, 8), (130, 772, 1004), (256, 1152, 100), (33, 32, 4), (100, 20, 12)]:
    z = torch.randn(B, Dp, device=dev); Wc = torch.randn(C, Dp, device=dev) * 0.1; bc = torch.randn(C, device=dev)
    dl = torch.randn(B, C, device=dev)
    y = F_.linear_forward(z, Wc, bc)
    ref = (z.double() @ Wc.double().t() + bc.double())
    e1 = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    dWc = torch.empty_like(Wc); dbc = torch.empty(C, device=dev)
    dz = F_.linear_backward(dl, z, Wc, True, dWc, dbc, False)
    dz = dz[0] if isinstance(dz, (tuple, list)) else dz
    rdz = dl.double() @ Wc.double(); rdW = dl.double().t() @ z.double()
    e2 = (dz.double() - rdz).abs().max().item() / rdz.abs().max().item() if dz is not None else -1
    e3 = (dWc.double() - rdW).abs().max().item() / rdW.abs().max().item()
    flag = max(e1, e2, e3) > 2e-6
    bad += flag
    print(B, Dp, C, f"{e1:.2e} {e2:.2e} {e3:.2e}", "BAD" if flag else "")
print("bad", bad)
