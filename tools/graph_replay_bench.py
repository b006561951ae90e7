#!/usr/bin/env python3
"""Eager steps against hipGraph replays of the same captured step (tests/test_gpu_graph_capture.py): device time per step.
usage (GPU box): python tools/graph_replay_bench.py [B N D Q]"""
import sys, os, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_probing_amd import probe_heads
from efficient_probing_amd.engine import ProbeHeadEngine

B, Nn, D, Q = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 256, 768, 8)
C = 1000
dev = "cuda:0"


class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.head = torch.nn.Linear(D, C)


torch.manual_seed(0)
e = Enc()
probe_heads.build_probe_head(e, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
eng = ProbeHeadEngine(e.head.to(dev).train(), optimizer="lars", lr=0.4)
g = torch.Generator(device=dev).manual_seed(1)
xs = [torch.randn(B, Nn, D, device=dev, generator=g) for _ in range(4)]
t = torch.randint(0, C, (B,), device=dev, generator=g)
for i in range(60):
    eng.train_step(xs[i % 4], t, lr=0.4)
torch.cuda.synchronize()


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


graphs = []
side = torch.cuda.Stream()
for k in range(4):
    gr = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            eng.train_step(xs[k], t, lr=0.4)
    torch.cuda.current_stream().wait_stream(side)
    graphs.append(gr)
for rep in range(3):
    te = timed(lambda i: eng.train_step(xs[i % 4], t, lr=0.4), 200)
    tg = timed(lambda i: graphs[i % 4].replay(), 200)
    print(f"{B}x{Nn}x{D} q{Q}: eager {te:.4f} ms/step, graph replay {tg:.4f} ms/step")
