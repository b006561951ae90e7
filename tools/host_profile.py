#!/usr/bin/env python3
"""Where the Python time of one engine.train_step() goes (GPU box).  A small batch makes the device faster than the host, so the
wall time per call IS the host cost; cProfile then splits it.  usage: host_profile.py [iters] [B]   (EP_FAST_STEP=0: the general path)"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from efficient_probing_amd import probe_heads
from efficient_probing_amd.engine import ProbeHeadEngine

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
N, D, Q, C = 64, 768, 8, 1000


class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.head = torch.nn.Linear(D, C)


torch.manual_seed(0)
enc = Enc()
probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C, num_heads=16, model="vit_base_patch16"))
head = enc.head.to(dev).train()
eng = ProbeHeadEngine(head, optimizer="lars", lr=0.1)
xs = [torch.randn(B, N, D, device=dev) for _ in range(4)]
ts = [torch.randint(0, C, (B,), device=dev) for _ in range(4)]
for i in range(50):
    eng.train_step(xs[i % 4], ts[i % 4], lr=0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n_it):
    eng.train_step(xs[i % 4], ts[i % 4], lr=0.1)
host = (time.perf_counter() - t0) / n_it
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n_it
print(f"train_step: {host * 1e6:.1f} us of host time per call ({total * 1e6:.1f} us per call with the device drained; B = {B}, fast path {'on' if eng._fast_ok else 'off'})")
pr = cProfile.Profile()
pr.enable()
for i in range(n_it):
    eng.train_step(xs[i % 4], ts[i % 4], lr=0.1)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:32]))
