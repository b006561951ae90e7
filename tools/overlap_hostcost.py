#!/usr/bin/env python3
"""One-rank check of the pipelined data-parallel schedule's host cost (GPU box): ms/step with and without it."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from argparse import Namespace
from efficient_probing_amd import probe_heads
from efficient_probing_amd.engine import ProbeHeadEngine
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
if "--no-dist" not in sys.argv:
    dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
B, Nn, D, Q, C = 1024, 256, 768, 8, 1000
res = {}
for mode in (False, "force"):
    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.head = torch.nn.Linear(D, C)
    torch.manual_seed(0); enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
    eng = ProbeHeadEngine(enc.head.to(dev).train(), optimizer="lars", lr=0.4, overlap_comm=mode)
    xs = [torch.randn(B, Nn, D, device=dev) for _ in range(4)]; ts = [torch.randint(0, C, (B,), device=dev) for _ in range(4)]
    for i in range(10): eng.train_step(xs[i % 4], ts[i % 4])
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200): eng.train_step(xs[i % 4], ts[i % 4])
    host = time.perf_counter() - t0
    eng.flush(); torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    res[str(mode)] = {"ms_per_step": round(tot / 200 * 1e3, 4), "host_enqueue_ms_per_step": round(host / 200 * 1e3, 4)}
res["dist"] = dist.is_initialized()
print(json.dumps(res))
if dist.is_initialized():
    dist.destroy_process_group()
