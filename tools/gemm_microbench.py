#!/usr/bin/env python3
"""Times the six contractions of the head tail through the C ABI (GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_
B, D, Q, C = int(os.environ.get("B", 1024)), 768, 8, 1000
dev = "cuda:0"
P = torch.randn(B, Q, D, device=dev); Wv = torch.randn(D, D, device=dev) * 0.03
z = torch.randn(B, D, device=dev); Wc = torch.randn(C, D, device=dev) * 0.03; bc = torch.randn(C, device=dev)
dl = torch.randn(B, C, device=dev); dy = torch.randn(B, D, device=dev)
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / it, 1)
r = {}
r["project_fwd"] = timeit(lambda: F_.project_forward(P, Wv))
r["logits"] = timeit(lambda: F_.linear_forward(z, Wc, bc))
dWc = torch.empty_like(Wc); dbc = torch.empty(C, device=dev)
r["lin_bwd(dz+dWc+dbc)"] = timeit(lambda: F_.linear_backward(dl, z, Wc, True, dWc, dbc, False))
dWv = torch.empty_like(Wv)
r["proj_bwd(dP+dWv)"] = timeit(lambda: F_.project_backward(dy, None, P, Wv, None, True, dWv, False, True))
r["dP_only"] = timeit(lambda: F_.project_backward(dy, None, P, Wv, None, True, None, False, False))
r["torch_mm_logits"] = timeit(lambda: torch.addmm(bc, z, Wc.t()))
r["env"] = {k: v for k, v in os.environ.items() if k.startswith("EP_")}
print(json.dumps(r))
