#!/usr/bin/env python3
"""A/B micro-benchmark of the EP pooling kernels alone (GPU box).  Every launch is bracketed by its own
event pair and all buffers are preallocated, so the figures are device durations (not host launch rate).
Environment knobs read by the library: EP_POOL_WG_PER_CU, EP_POOL_GRID, EP_POOL_ABLATE (diagnostics only)."""
import argparse, os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import _native as N_

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=256)
ap.add_argument("--D", type=int, default=768); ap.add_argument("--Q", type=int, default=8)
ap.add_argument("--iters", type=int, default=30); ap.add_argument("--bufs", type=int, default=1)
ap.add_argument("--bwd", action="store_true"); ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = N_.load()
xs = [torch.randn(a.B, a.N, a.D, device=dev).to(torch.bfloat16 if a.bf16 else torch.float32) for _ in range(a.bufs)]
DT = N_.EP_DTYPE_BF16 if a.bf16 else N_.EP_DTYPE_F32
cls = torch.randn(a.Q, a.D, device=dev) * 0.02
scale = a.D ** -0.5
P = torch.empty(a.B, a.Q, a.D, device=dev); S = torch.empty(a.B, a.Q, a.N, device=dev)
ML = torch.empty(a.B, a.Q, 4, device=dev); dP = torch.randn(a.B, a.Q, a.D, device=dev)
dcls = torch.empty(a.Q, a.D, device=dev)
nws = lib.ep_pool_workspace_bytes(a.B, a.N, a.D, a.Q)
ws = torch.empty(nws, device=dev, dtype=torch.uint8)
st = N_.current_stream_ptr(dev)

def fwd(i):
    x = xs[i % a.bufs]
    N_.check(lib.ep_pool_forward(x.data_ptr(), DT, a.N * a.D, 0, a.B, a.N, a.D, cls.data_ptr(), 0, a.Q,
                                 scale, P.data_ptr(), S.data_ptr(), ML.data_ptr(), ws.data_ptr(), nws, st), "fwd")
def bwd(i):
    x = xs[i % a.bufs]
    N_.check(lib.ep_pool_backward(x.data_ptr(), DT, a.N * a.D, 0, a.B, a.N, a.D, a.Q, scale, S.data_ptr(),
                                  ML.data_ptr(), dP.data_ptr(), dcls.data_ptr(), 0, ws.data_ptr(), nws, st), "bwd")
def timeit(fn):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for i, (e0, e1) in enumerate(ev):
        e0.record(); fn(i); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return statistics.median(ts), ts[0]

res = {"B": a.B}
gb = a.B * a.N * a.D * (2 if a.bf16 else 4) / 1e3
t, tmin = timeit(fwd)
res.update(fwd_us=round(t, 1), fwd_min=round(tmin, 1), fwd_GBs=round(gb / t, 1), fwd_kernel=N_.pool_kernel_name(a.B, a.N, a.D, a.Q, False)
           if hasattr(N_, "pool_kernel_name") else "")
if a.bwd:
    ML[:, :, 2] = 0
    t, tmin = timeit(bwd)
    res.update(bwd_us=round(t, 1), bwd_min=round(tmin, 1), bwd_GBs=round(gb / t, 1))
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("EP_")}
print(json.dumps(res))
