#!/usr/bin/env python3
"""Host-side cost of one iteration of the loop surface (GPU box): python time per engine.train_step() call while the device
queue runs ahead, and per iteration of engine_finetune.train_one_epoch with the step stubbed out.  usage: host_overhead.py [iters]"""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from efficient_probing_amd import probe_heads, engine_finetune as EF
from efficient_probing_amd.token_store import ResidentTokenStore
from efficient_probing_amd.util.lars import LARS

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
B, N, D, Q, C = 1024, 256, 768, 8, 1000


class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.head = torch.nn.Linear(D, C)


torch.manual_seed(0)
enc = Enc()
probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C, num_heads=16, model="vit_base_patch16"))
enc.to(dev)
x = torch.randn(2 * B, N, D, device=dev); t = torch.randint(0, C, (2 * B,), device=dev)
store = ResidentTokenStore.from_tensors(x, t)


class Epochs:
    def __len__(self): return n_it
    def __iter__(self):
        k, ep = 0, 0
        while k < n_it:
            for bt in store.batches(B, epoch=ep):
                if k == n_it: return
                k += 1
                yield bt
            ep += 1


opt = LARS(enc.head.parameters(), lr=0.1, weight_decay=0.0)
a = Namespace(lr=0.1, min_lr=0.0, warmup_epochs=10, epochs=90, accum_iter=1, amp="none")
crit = torch.nn.CrossEntropyLoss()
with contextlib.redirect_stdout(open(os.devnull, "w")):
    EF.train_one_epoch(enc, crit, Epochs(), opt, dev, 0, None, args=a)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); EF.train_one_epoch(enc, crit, Epochs(), opt, dev, 1, None, args=a); torch.cuda.synchronize()
    full = (time.perf_counter() - t0) / n_it
eng = EF.get_engine(enc, opt, a)
bt = next(iter(store.batches(B, epoch=0)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_it): eng.train_step(bt[0], bt[2], lr=0.1, image_index=bt[1])
host_step = (time.perf_counter() - t0) / n_it
torch.cuda.synchronize()
dev_step = (time.perf_counter() - t0) / n_it
real = eng.train_step
eng.train_step = lambda *aa, **kk: None
eng.read_stats = lambda: (1.0, 1.0, 1.0, 0.0)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    t0 = time.perf_counter(); EF.train_one_epoch(enc, crit, Epochs(), opt, dev, 2, None, args=a)
    loop_only = (time.perf_counter() - t0) / n_it
eng.train_step = real
print(f"per iteration: train_one_epoch {full * 1e6:.0f} us | device-bound raw loop {dev_step * 1e6:.0f} us | "
      f"python inside engine.train_step {host_step * 1e6:.0f} us | loop without the step {loop_only * 1e6:.0f} us")
if os.environ.get("EP_HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        pr.enable(); EF.train_one_epoch(enc, crit, Epochs(), opt, dev, 3, None, args=a); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
