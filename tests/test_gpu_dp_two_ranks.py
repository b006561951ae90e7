"""Two data-parallel ranks on ONE GPU (gloo backend moving CUDA tensors): the real kernels, the real two-bucket
all-reduce and the pipelined schedule of engine.ProbeHeadEngine, against an in-process simulation of the same two ranks
(plain schedule, gradients summed by hand).  Every rank must end with the same parameters, bit for bit equal to the
simulation.  Needs an MI355X (pytest -m gpu)."""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = dict(B=24, N=40, D=256, Q=8, C=30)
STEPS, LRS = 4, [0.05, 0.3, 0.2, 0.1]


def _build(overlap_comm, world_aware=True):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(SHAPE["D"], SHAPE["C"])
    torch.manual_seed(0)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=SHAPE["Q"], d_out=1, nb_classes=SHAPE["C"]))
    return ProbeHeadEngine(enc.head.to("cuda:0").train(), optimizer="lars", weight_decay=1e-3, overlap_comm=overlap_comm)


def _data(rank, step):
    g = torch.Generator().manual_seed(1000 * rank + step)
    x = torch.randn(SHAPE["B"], SHAPE["N"], SHAPE["D"], generator=g)
    t = torch.randint(0, SHAPE["C"], (SHAPE["B"],), generator=g)
    return x.to("cuda:0"), t.to("cuda:0")


def _worker(rank, world, port, out_dir, overlap):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    eng = _build(overlap_comm=overlap)            # None: the engine's default = the plain single all-reduce
    assert eng.world == world and eng._pipelined == bool(overlap)
    for step in range(STEPS):
        x, t = _data(rank, step)
        eng.train_step(x, t, lr=LRS[step])
    eng.flush()
    eng.sync_buffers()
    torch.cuda.synchronize()
    torch.save({"p": eng.flat_p.cpu(), "mu": eng.state[0].cpu(), "rm": eng.bn.running_mean.cpu()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [None, True], ids=["default_single_allreduce", "opt_in_two_bucket_overlap"])
def test_two_ranks_equal_simulation(tmp_path, overlap):
    import torch.multiprocessing as mp
    world, port = 2, 29650 + (os.getpid() % 200) + (300 if overlap else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), overlap), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(got[0]["p"], got[1]["p"]) and torch.equal(got[0]["mu"], got[1]["mu"])
    assert torch.equal(got[0]["rm"], got[1]["rm"])                       # sync_buffers: rank 0's running statistics
    # in-process simulation: two replicas, plain schedule, gradients summed by hand, 1/world in the optimizer
    reps = [_build(overlap_comm=False) for _ in range(world)]
    for e in reps:
        e.world = world                                                  # inv_scale = 1 / (loss_scale * world)
    for step in range(STEPS):
        for r, e in enumerate(reps):
            x, t = _data(r, step)
            e.forward_backward(x, t)
        total = reps[0].flat_g + reps[1].flat_g
        for e in reps:
            e.flat_g.copy_(total)
            e.optimizer_step(LRS[step])
    assert torch.equal(reps[0].flat_p, reps[1].flat_p)
    assert torch.equal(got[0]["p"], reps[0].flat_p.cpu())
    assert torch.equal(got[0]["mu"], reps[0].state[0].cpu())
    assert torch.equal(got[0]["rm"], reps[0].bn.running_mean.cpu())


def test_bench_two_ranks_end_to_end_on_one_device():
    """``python bench.py --gpus 2`` end to end with the real kernels: the launcher spawns two fresh child ranks, they
    rendezvous on 127.0.0.1, run the data-parallel step (flat-gradient all-reduce, 1 / world in the optimizer), barrier,
    flush(), take the MAX of the elapsed time and rank 0 prints the contract line.  On this one-GPU box both ranks share
    device 0 and the collectives go over gloo (EP_BENCH_SHARE_DEVICE=1); on a node the same code runs over RCCL."""
    import json
    import subprocess
    env = dict(os.environ, EP_BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--spinup", "4",
           "--batch", "256", "--no-cpu-baseline", "--no-bf16-secondary", "--no-north-star", "--no-configs",
           "--no-through-engine", "--kernel-iters", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 6
    assert line["config"]["global_batch"] == 512 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and np.isfinite(line["check"]["mean_loss_over_timed_steps"])
    assert line["value"] == pytest.approx(512 * 6 / (line["ms_per_step"] * 6e-3), rel=1e-3)   # whole-job images per second
    # round 4: the data-parallel object -- the ONE exchange step of the path timed alone, and both schedules in one invocation
    dp = line["data_parallel"]
    assert dp["allreduce_us"] > 0 and dp["allreduce_bytes"] == 4 * (8 * 768 + 768 * 768 + 1000 * 768 + 1000)
    assert dp["overlap_comm"]["on"]["pipelined"] is True and dp["overlap_comm"]["on"]["value"] > 0
    assert dp["overlap_comm"]["off"]["value"] == pytest.approx(line["value"], rel=1e-6)
    assert np.isfinite(dp["overlap_comm"]["on"]["mean_loss_over_timed_steps"])
    assert line["scaling"] == "weak"
    # the protocol's fixed global batch split over the ranks: strong scaling, lr from the GLOBAL batch
    cmd2 = [c for c in cmd if c not in ("--batch", "256")] + ["--global-batch", "512"]
    r = subprocess.run(cmd2, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line2 = json.loads(r.stdout.strip().splitlines()[-1])
    assert line2["scaling"] == "strong" and line2["config"]["batch_per_gpu"] == 256 and line2["config"]["global_batch"] == 512
