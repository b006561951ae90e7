#!/usr/bin/env python3
"""Round 5: the single-read 17 .. 32-query token passes (csrc/ep_pool_mm2.hip on fp32 tokens, the two-block form of
csrc/ep_pool_mb.hip on bf16 tokens) against a float64 evaluation on the GPU, and their device durations.
usage: python tools/q32_check.py [--time] [--bf16]"""
import argparse, os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_probing_amd import functional as F_, _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--time", action="store_true"); ap.add_argument("--bf16", action="store_true")
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=256)
ap.add_argument("--D", type=int, default=768); ap.add_argument("--Q", type=int, default=32)
a = ap.parse_args()
lib = N.load()
dev = "cuda:0"


def ref64(x, cls, scale, dP, delta):
    xb = x.double()
    s = torch.matmul((cls * scale).double(), xb.transpose(1, 2))
    A = torch.softmax(s, -1)
    P = torch.matmul(A, xb)
    dA = torch.matmul(dP.double(), xb.transpose(1, 2))
    dcls = scale * torch.matmul(A * (dA - delta), xb).sum(0)
    return P, s, dcls


bad = 0
SHAPES = [(5, 256, 768, 32, 1.0), (5, 256, 768, 32, 30.0), (7, 197, 768, 32, 8.0), (9, 50, 256, 17, 8.0), (300, 77, 384, 24, 8.0),
          (6, 33, 512, 31, 8.0), (4, 16, 640, 32, 8.0), (4, 1, 896, 20, 8.0), (70, 196, 1024, 32, 8.0), (3, 65, 1152, 32, 8.0), (3, 15, 768, 16, 8.0)]
for (B, Nn, D, Q, amp) in SHAPES:
    for storage in ("f32", "bf16"):
        g = torch.Generator(device=dev).manual_seed(B + Nn + D + Q)
        x = torch.randn(B, Nn, D, device=dev, generator=g)
        if storage == "bf16":
            x = x.to(torch.bfloat16)
        cls = torch.randn(Q, D, device=dev, generator=g) * amp / D ** 0.5
        dP = torch.randn(B, Q, D, device=dev, generator=g)
        names = [lib.ep_pool_kernel_name_ex(B, Nn, D, Q, b, 1 if storage == "bf16" else 0).decode() for b in (0, 1)]
        P, S, ML = F_.pool_forward(x, cls, 1.0)
        ML2 = ML.clone(); ML2[:, :, 2] = 0.25
        dcls = F_.pool_backward(x, S, ML2, dP, 1.0)
        Pr, Sr, dr = ref64(x.float(), cls, 1.0, dP, 0.25)
        eP = (P.double() - Pr).abs().max().item() / Pr.abs().max().item()
        eS = (S.double() - Sr).abs().max().item() / Sr.abs().max().item()
        eG = (dcls.double() - dr).abs().max().item() / dr.abs().max().item()
        ok = eP < 2e-6 and eS < 2e-6 and eG < 2e-5
        bad += not ok
        print(f"{B:4d} {Nn:4d} {D:5d} q{Q:2d} amp {amp:4.1f} {storage} {names[0]:28s} {names[1]:28s} P {eP:.1e} S {eS:.1e} dcls {eG:.1e} {'ok' if ok else 'BAD'}")
print("ALL OK" if not bad else f"{bad} BAD")

if a.time:
    B, Nn, D, Q = a.B, a.N, a.D, a.Q
    DT = N.EP_DTYPE_BF16 if a.bf16 else N.EP_DTYPE_F32
    xs = [torch.randn(B, Nn, D, device=dev).to(torch.bfloat16 if a.bf16 else torch.float32) for _ in range(3)]
    cls = torch.randn(Q, D, device=dev) * 0.02
    P = torch.empty(B, Q, D, device=dev); S = torch.empty(B, Q, Nn, device=dev); ML = torch.zeros(B, Q, 4, device=dev)
    dP = torch.randn(B, Q, D, device=dev); dcls = torch.empty(Q, D, device=dev)
    nws = lib.ep_pool_workspace_bytes(B, Nn, D, Q); ws = torch.zeros(nws, device=dev, dtype=torch.uint8)
    st = N.current_stream_ptr(torch.device(dev))
    sc = D ** -0.5
    def fwd(i):
        N.check(lib.ep_pool_forward(xs[i % 3].data_ptr(), DT, Nn * D, 0, B, Nn, D, cls.data_ptr(), 0, Q, sc, P.data_ptr(), S.data_ptr(), ML.data_ptr(), ws.data_ptr(), nws, st), "fwd")
    def bwd(i):
        N.check(lib.ep_pool_backward(xs[i % 3].data_ptr(), DT, Nn * D, 0, B, Nn, D, Q, sc, S.data_ptr(), ML.data_ptr(), dP.data_ptr(), dcls.data_ptr(), 0, ws.data_ptr(), nws, st), "bwd")
    def timeit(fn, iters=30):
        for i in range(5): fn(i)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (e0, e1) in enumerate(ev):
            e0.record(); fn(i); e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return round(statistics.median(ts), 1), round(ts[0], 1)
    gb = B * Nn * D * (2 if a.bf16 else 4) / 1e3
    f, fm = timeit(fwd); b, bm = timeit(bwd)
    print(json.dumps(dict(B=B, N=Nn, D=D, Q=Q, bf16=a.bf16, fwd_us=f, fwd_min=fm, fwd_GBs=round(gb / f), bwd_us=b, bwd_min=bm, bwd_GBs=round(gb / b),
                          fwd_kernel=lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 0, int(a.bf16)).decode(), bwd_kernel=lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 1, int(a.bf16)).decode(),
                          env={k: v for k, v in os.environ.items() if k.startswith("EP_")})))
