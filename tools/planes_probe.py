"""Times and checks the planes contraction (csrc/ep_planes.hip, ep_planes_big.hip) stand-alone: C = A (M x K) W^T (N x K) against
pre-split planes, error against float64, event-timed median.  Run once per EP_PLANES_BIG setting (read once per process):
  EP_PLANES_BIG=0 python tools/planes_probe.py ; EP_PLANES_BIG=1 python tools/planes_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_probing_amd import functional as F_  # noqa: E402

DEV = "cuda:0"
SHAPES = [(1024, 4096, 4096), (1024, 4096, 1000), (1024, 1000, 4096), (8192, 4096, 512), (8192, 512, 4096), (1024, 768, 768),
          (1000, 776, 333), (130, 100, 250), (64, 32, 64), (1025, 132, 4096), (257, 4000, 129), (300, 1000, 4096), (128, 64, 128), (4096, 1152, 1152)]


if os.environ.get("PROBE_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["PROBE_SHAPES"].split(",")]


def main():
    print(f"# EP_PLANES_BIG={os.environ.get('EP_PLANES_BIG', '(unset)')}  M x K x N | us | TFLOP/s (fp32-equivalent) | err vs float64 | err of the f32 kernel")
    for M, K, N in SHAPES:
        g = torch.Generator().manual_seed(M + K + N)
        A = torch.randn(M, K, generator=g).to(DEV)
        W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        pn, _ = F_.planes_split(W)
        got = F_.matmul_planes(A, pn, N, bias=b)
        ref = A.double() @ W.double().t() + b.double()
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        f32 = F_.linear_forward(A, W, b)
        err32 = float((f32.double() - ref).abs().max() / ref.abs().max())
        for _ in range(5):
            F_.matmul_planes(A, pn, N, bias=b)
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); F_.matmul_planes(A, pn, N, bias=b); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        flag = "" if (err < 3e-6 and err < 3 * err32 + 2e-7) else "   <-- BAD"
        print(f"{M:5d} x {K:5d} x {N:5d} | {us:8.1f} | {2.0 * M * N * K / us * 1e-6:7.1f} | {err:.2e} | {err32:.2e}{flag}", flush=True)


if __name__ == "__main__":
    main()
