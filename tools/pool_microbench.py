#!/usr/bin/env python3
"""A/B micro-benchmark of the EP pooling kernels alone (GPU box).  Environment knobs read by the
library: EP_POOL_NSLOT, EP_POOL_WG_PER_CU, EP_POOL_ABLATE (diagnostics only)."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_, _native as N_

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=256)
ap.add_argument("--D", type=int, default=768); ap.add_argument("--Q", type=int, default=8)
ap.add_argument("--iters", type=int, default=20); ap.add_argument("--bufs", type=int, default=3)
ap.add_argument("--bwd", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
xs = [torch.randn(a.B, a.N, a.D, device=dev) for _ in range(a.bufs)]
cls = torch.randn(a.Q, a.D, device=dev) * 0.02
scale = a.D ** -0.5
P, S, ML = F_.pool_forward(xs[0], cls, scale)
torch.cuda.synchronize()
def timeit(fn):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / a.iters
res = {}
t = timeit(lambda i: F_.pool_forward(xs[i % a.bufs], cls, scale))
gb = a.B * a.N * a.D * 4 / 1e9
res["fwd_us"] = round(t * 1e6, 1); res["fwd_GBs"] = round(gb / t, 1)
if a.bwd:
    dP = torch.randn_like(P); ML[:, :, 2] = 0
    t = timeit(lambda i: F_.pool_backward(xs[i % a.bufs], S, ML, dP, scale))
    res["bwd_us"] = round(t * 1e6, 1); res["bwd_GBs"] = round(gb / t, 1)
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("EP_POOL")}
print(json.dumps(res))
