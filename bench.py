#!/usr/bin/env python3
"""EP-head throughput benchmark (BASELINE.json metric: EP-head images/sec).

    python bench.py --gpus N --steps K --warmup W

A "step" is one full training iteration of the probe head on one batch of synthetic tokens
already resident in HBM: EP pooling forward, per-query value projection, BatchNorm, classifier,
cross-entropy, the whole backward, [one RCCL all-reduce of the flat gradients when N > 1], LARS
update.  Default workload: BASELINE.json configs[1] (DINOv2 ViT-B/14 tokens 256x768, Q=8,
1000 classes); ``--workload ns`` is the north-star shape (ViT-B/16, 197x768).

One JSON line is printed by rank 0 (see the driver contract in the task statement) with two extra
objects: ``roofline`` for the dominant kernel of the step AS IT RUNS IN THE STEP (the second token pass, HBM-bound: it
carries the in-pass dP tasks and the weight-gradient side workgroups; bracketed by the library's own HIP events; the first
pass and both passes launched alone are sub-objects), and ``cpu_baseline`` = the op-for-op torch-CPU port of the reference step
(oracle/torch_port.py) timed on this box's host cores for a bounded ~15 s sample.  The default EP run at N = 1 adds
``bf16_token_storage``: the same step on the same tokens stored as bf16 (fp32 arithmetic and results) -- a secondary
figure, never ``value``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N tokens, D, Q, classes, description)
    "c2": (256, 768, 8, 1000, "DINOv2 ViT-B/14 tokens 256x768, EP q=8, 1000 classes (BASELINE configs[1])"),
    "ns": (197, 768, 8, 1000, "ViT-B/16 tokens 197x768, EP q=8, 1000 classes (north-star shape)"),
    "c1": (196, 384, 1, 100, "DINO ViT-S/16 tokens 196x384, EP q=1, 100 classes (BASELINE configs[0])"),
    "c3": (196, 1024, 8, 1000, "MAE ViT-L/16 tokens 196x1024, EP q=8 (BASELINE configs[2])"),
    "c4": (256, 1152, 8, 1000, "SigLIP2 SO400M/14 tokens 256x1152, EP q=8 (BASELINE configs[3])"),
    "c5": (196, 4096, 8, 1000, "DINOv3 ViT-7B/16 tokens 196x4096, EP q=8 (BASELINE configs[4])"),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--head", default="ep", choices=["ep", "coca", "siglip", "cae", "jepa", "aim", "simpool", "esimpool", "cait", "clip", "dolg", "cbam", "dinovit", "abmilp"],
                    help="probe head: ep (the headline), the CoCa attentional pooler or the SigLIP attention-pool head on "
                         "the same token passes, or the matrix-core-bound AbMILP head")
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU per step (weak scaling); default 1024 (256 for --head abmilp / dolg / dinovit)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling at the published protocol's batch (reference README.md:119-120: effective batch 4096): the "
                         "per-GPU batch is G / --gpus and lr = 0.1 * G / 256 (main_linprobe.py:572-573) whatever N is; the line then "
                         "says \"scaling\": \"strong\"")
    ap.add_argument("--queries", type=int, default=None,
                    help="EP head: number of learned queries (default: the workload's, 8 for the BASELINE configurations; the "
                         "published accuracy rows of the reference use --ep_queries 32, README.md:133-134)")
    ap.add_argument("--spinup", type=int, default=40,
                    help="untimed steps run during setup, before the W warm-up steps, so that the chip's clock has settled under load")
    ap.add_argument("--mark-every", type=int, default=10, help="steps between the device events the step-time spread is read from")
    ap.add_argument("--buffers", type=int, default=4, help="distinct token buffers rotated through (HBM, not cache)")
    ap.add_argument("--tokens", default="f32", choices=["f32", "bf16"],
                    help="storage type of the tokens in HBM (arithmetic is fp32 either way)")
    ap.add_argument("--arith", default="fp32", choices=["fp32", "bf16_autocast"],
                    help="arithmetic of the EP step's six contractions (ep_head_step.arith): fp32 (default, the headline) or the "
                         "AMP-bf16 mode of the published --amp bfloat16 runs (one bf16 product, fp32 accumulation) -- an explicit "
                         "secondary mode: the line says so in `dtype` and `config`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-secondary", action="store_true", help="skip the bf16-token-storage line of the default EP run")
    ap.add_argument("--no-north-star", action="store_true", help="skip the north-star-shape (197x768) object of the default EP run")
    ap.add_argument("--no-configs", action="store_true", help="skip the 20-step secondaries of the other BASELINE configurations (c1, c3, c4, c5; CoCa and AbMILP at c4)")
    ap.add_argument("--no-through-engine", action="store_true",
                    help="skip the train_one_epoch() secondary (the same workload through the reference's loop surface)")
    ap.add_argument("--engine-steps", type=int, default=200, help="iterations of the train_one_epoch() secondary")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check without a GPU: spawn / join the ranks (gloo), all-reduce a flat buffer of the "
                         "gradient's size once and print the rank count -- no kernels, no throughput")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--kernel-iters", type=int, default=20)
    return ap.parse_args()


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh child processes, one per GPU (what the reference
    gets from ``torchrun`` / main_linprobe.py:581-583 + util/misc.py:214-257), BEFORE this process touches the GPU;
    the children rendezvous on 127.0.0.1 and rank 0 prints the one JSON line.  Returns the worst child exit code."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll all ranks: the first one that dies takes the others with it (they would otherwise sit in the rendezvous or in an
    # all-reduce until the collective's timeout); children never outlive this process
    import time
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for pr in list(live):
                code = pr.poll()
                if code is not None:
                    live.remove(pr)
                    rc = max(rc, abs(code))
            if live and rc == 0:
                time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    return rc


def rendezvous_only(args, world, rank):
    """The launcher / rendezvous / collective plumbing on CPU (gloo): what ``tests/test_bench_launcher.py`` runs."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    Nn, D, Q, Cc, _ = WORKLOADS[args.workload]
    flat = torch.full((Q * D + D * D + Cc * D + Cc,), float(rank + 1))
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        ranks = dist.get_world_size()
    else:
        ranks = 1
    ok = bool((flat == world * (world + 1) / 2).all())
    if args.global_batch and args.global_batch % world != 0:
        raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of --gpus {world}")
    bpg = args.global_batch // world if args.global_batch else (args.batch or 1024)
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "rccl_ranks": ranks, "backend": "gloo",
                          "allreduce_elements": flat.numel(), "allreduce_ok": ok, "batch_per_gpu": bpg,
                          "global_batch": bpg * world, "lr": 0.1 * bpg * world / 256,
                          "scaling": "strong" if args.global_batch else "weak"}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok and ranks == args.gpus else 3


F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32: 256 FLOP/cycle/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md)
BF16X3_PEAK_TFLOPS = 2500.0 / 6  # fp32-accurate products on the bf16 pipe: six of the 2.5 PFLOP/s dense bf16 products each


def bench_abmilp(args, torch, dist, dev, world, rank, Nn, D, Cc, desc, B):
    """--head abmilp / dolg / dinovit: the matrix-core-bound heads (f4).  Same contract line; the roofline
    object prices the dominant kernel (AbMILP: the qkv projection; DOLG: the 1x1 convolution over all token rows) against
    the dense fp32 MFMA peak."""
    dolg, dino = args.head == "dolg", args.head == "dinovit"
    label = {"dolg": "DOLG", "dinovit": "DINOv2-block", "abmilp": "AbMILP"}[args.head]
    from argparse import Namespace
    from efficient_probing_amd import probe_heads, functional as F_
    from efficient_probing_amd.engine import make_engine

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.patch_embed = Namespace(num_patches=Nn)
            self.head = torch.nn.Linear(D, Cc)
    torch.manual_seed(0)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features=args.head, nb_classes=Cc, abmilp_sa="both", abmilp_act="tanh",
                                                abmilp_depth=2, abmilp_cond=None, abmilp_content="all"))
    head = enc.head.to(dev).train()
    if args.arith != "fp32" and args.head != "abmilp":
        raise SystemExit("--arith bf16_autocast is implemented for the EP, CoCa and AbMILP heads only")
    eng = make_engine(head, optimizer="lars", lr=0.1 * (B * world) / 256, weight_decay=0.0,
                      **({"arithmetic": args.arith} if args.arith != "fp32" else {}))
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    nbuf = min(args.buffers, 2)
    xs = [torch.randn(B, Nn, D, device=dev, generator=gen) for _ in range(nbuf)]
    ts = [torch.randint(0, Cc, (B,), device=dev, generator=gen) for _ in range(nbuf)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    steps, warmup = args.steps, args.warmup
    for i in range(min(args.spinup, 10) + warmup):        # (these steps take milliseconds: 10 of them settle the clock)
        eng.train_step(xs[i % nbuf], ts[i % nbuf])
    eng.read_stats()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(xs[i % nbuf], ts[i % nbuf])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_sum, _, _, bad = eng.read_stats()
    # dominant kernel alone: qkv = x Wqkv^T over all B*N token rows (HIP events on the launch stream)
    xf = xs[0].view(B * Nn, D)
    Wqkv = (head[0].conv1.weight.detach().view(D, D) if dolg else head[0].dino_block.attn.qkv.weight.detach() if dino
            else head[0].self_attn.qkv.weight.detach())
    F_.linear_forward(xf, Wqkv, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.kernel_iters):
        F_.linear_forward(xf, Wqkv, None)
    e1.record()
    torch.cuda.synchronize()
    t_k = e0.elapsed_time(e1) * 1e-3 / args.kernel_iters
    k_flop = 2.0 * B * Nn * D * (D if dolg else 3 * D)
    # which kernel that was, and the peak it is priced against: the exact-f32 instruction (157 TFLOP/s) or -- large
    # contractions since round 4 -- the bf16 x3 tile (csrc/ep_wgrad3.h): six bf16 MFMAs per 16 x 16 x 32 block, i.e. a sixth of
    # the 2.5 PFLOP/s dense bf16 peak in fp32-equivalent FLOP
    kname = eng.lib.ep_linear_kernel_name(B * Nn, D if dolg else 3 * D, D).decode()
    k_peak = BF16X3_PEAK_TFLOPS if "b3" in kname else F32_MFMA_PEAK_TFLOPS
    # fwd + bwd, no token gradient (DOLG: the 1x1 convolution and its weight gradient)
    # DINOv2 block: 24 D^2 FLOP per token forward in the four projections; backward 32 D^2 (weight and input gradients of
    # qkv, proj and fc1 -- the fc2 ones collapse to per-image rows under the token mean, ep_dinovit.hip); attention 4 N^2 D
    # forward and 8 N^2 D backward per image
    step_flop_img = (4.0 * Nn * D * D if dolg else 56.0 * Nn * D * D + 12.0 * Nn * Nn * D if dino
                     else 24.0 * Nn * D * D + 12.0 * Nn * Nn * D + 2.0 * Nn * D)
    if rank == 0:
        value = B * world * steps / elapsed
        out = {
            "metric": label + "-head train images/sec", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True, "scaling": getattr(args, "_scaling", "weak"), "vs_baseline": None,
            "dtype": "f32" if args.arith == "fp32" else "bf16_autocast (contractions: one bf16 product, fp32 accumulation; softmax, tanh, BatchNorm, loss, optimizer f32)",
            "data": "synthetic",
            "config": {"workload": desc.split(",")[0] + (f", DOLG spatial attention (1x1 conv, BatchNorm2d, softplus), {Cc} classes" if dolg
                                                          else f", DINOv2 block (8-head self-attention + GELU MLP) and token mean, {Cc} classes" if dino
                                                          else f", AbMILP head (self-attention + tanh predictor), {Cc} classes"),
                       "tokens": Nn, "dim": D, "classes": Cc, "batch_per_gpu": B, "global_batch": B * world,
                       "optimizer": "lars", "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": kname + " (" + ("1x1 convolution" if dolg else "qkv projection") + ")",
                         "achieved": round(k_flop / t_k / 1e12, 1), "peak": k_peak, "unit": "TFLOP/s",
                         "peak_note": ("bf16 x3 at fp32 accuracy: 2.5 PFLOP/s dense bf16 / 6 products per fp32 product" if "b3" in kname
                                       else "v_mfma_f32_16x16x4_f32"),
                         "frac": round(k_flop / t_k / 1e12 / k_peak, 4), "traffic": None,
                         "us_per_launch": round(t_k * 1e6, 1), "algorithmic_flop": k_flop,
                         "step_flop_per_image": step_flop_img,
                         "step_frac": round(value / world * step_flop_img / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)},
            "check": {"mean_loss_over_timed_steps": round(loss_sum / max(1, steps), 5), "nonfinite_rows": bad},
        }
        if args.head == "abmilp" and os.environ.get("EP_ABMILP_PLANES", "1") != "0":
            # the dominant-kernel figure above is the f32 kernel alone; in the step five of the weight contractions
            # (12 of its 24 N D^2 FLOP per image) run as bf16 x3 at fp32 accuracy (csrc/ep_planes.hip), whose fp32-equivalent
            # peak is 2.5 PFLOP/s / 6 = 417 TFLOP/s -- step_frac stays priced against the f32 peak and says so
            out["roofline"]["step_frac_note"] = ("priced against the f32 MFMA peak; half of the step's weight-contraction FLOP run "
                                                 "on the bf16 pipe as three-term splits (fp32-equivalent peak 417 TFLOP/s)")
        if world == 1 and not args.no_cpu_baseline:
            from oracle import torch_port, abmilp_oracle, dinovit_oracle, dolg_oracle
            mk = {"dolg": dolg_oracle, "dinovit": dinovit_oracle, "abmilp": abmilp_oracle}[args.head]
            r = torch_port.time_train_steps(8, Nn, D, 1, Cc, budget_s=args.cpu_seconds, threads=min(32, os.cpu_count() or 8),
                                            make=lambda: mk.make_head(D, Cc))
            out["cpu_baseline"] = {"value": round(r["value"], 2), "unit": "images/s", "cores": r["threads"], "kind": "port",
                                   "sample": f"{r['steps']} train steps of batch {r['batch']} ({r['seconds']:.1f} s), "
                                             f"torch-CPU restatement of the reference {label} step"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher (nothing below this line runs in the parent)
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.exit(rendezvous_only(args, world, rank))
    import torch
    import torch.distributed as dist
    from argparse import Namespace
    from efficient_probing_amd import probe_heads, functional as F_
    from efficient_probing_amd.engine import make_engine

    # EP_BENCH_SHARE_DEVICE=1 (tests on a one-GPU box only): every rank runs on device 0 and the collectives go over
    # gloo (it moves device tensors through the host) -- the whole world > 1 path of this file with the real kernels
    share = world > 1 and os.environ.get("EP_BENCH_SHARE_DEVICE") == "1"
    backend = "gloo" if share else "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0 if share else local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)    # "nccl" IS RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
    rccl_ranks = dist.get_world_size() if world > 1 else 1
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)

    Nn, D, Q, Cc, desc = WORKLOADS[args.workload]
    B = args.batch or (256 if args.head in ("abmilp", "dolg", "dinovit") else 1024)
    scaling = "weak"
    if args.global_batch:
        if args.batch:
            raise SystemExit("--global-batch and --batch are exclusive")
        if args.global_batch % world != 0:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of --gpus {world}")
        B = args.global_batch // world                     # total work fixed as N grows
        scaling = "strong"
    args._scaling = scaling
    if args.head in ("abmilp", "dolg", "dinovit"):
        return bench_abmilp(args, torch, dist, dev, world, rank, Nn, D, Cc, desc, B)
    if args.queries and args.head == "ep":
        Q = args.queries
        desc = desc.replace("EP q=8", f"EP q={Q}").replace("EP q=1", f"EP q={Q}")
    if args.head == "coca":
        Q = 8                                              # 8 query heads of image query 0 (coca_pytorch.py:259)
        desc = desc.split(",")[0] + f", CoCa pooler (8 heads x 64, 196 image queries), {Cc} classes"
    if args.head == "jepa":
        Q = 16                                             # --num_heads default (main_linprobe.py:116)
        desc = desc.split(",")[0] + f", V-JEPA attentive pooler (16 heads, LayerNorm-ed keys / values, MLP x4), {Cc} classes"
    if args.head in ("simpool", "esimpool"):
        Q = 1 if args.head == "simpool" else 12            # probe_heads.py:66-69
        desc = desc.split(",")[0] + (f", SimPool (mean-token query, LayerNorm-ed keys / values, wq / wk), {Cc} classes"
                                     if args.head == "simpool" else
                                     f", SimPool without linear maps (12 channel-slice heads), {Cc} classes")
    if args.head == "cbam":
        Q = 1
        if int(round(Nn ** 0.5)) ** 2 != Nn:
            raise SystemExit("--head cbam: the token count must be a perfect square")
        desc = desc.split(",")[0] + f", CBAM pooling (channel + 7x7 spatial gates), {Cc} classes"
    if args.head == "clip":
        Q = 4                                              # AttentionPool2d default heads (attention_pool2d.py:117)
        if Nn not in (196, 256):
            raise SystemExit("--head clip: the learned position embedding fixes the token count at 14 x 14 or 16 x 16")
        desc = desc.split(",")[0] + f", CLIP attention pooling (4 heads, mean-row query, position embedding), {Cc} classes"
    if args.head == "cait":
        Q = 4                                              # CAPooling default heads (other_pool.py:392)
        desc = desc.split(",")[0] + f", CaiT class-attention pooling (4 heads, LayerScale, MLP x4), {Cc} classes"
    if args.head == "aim":
        Q = 16                                             # --num_heads default (main_linprobe.py:116)
        desc = desc.split(",")[0] + f", AIM attention pooling (16 heads, batch-normalised tokens), {Cc} classes"
    if args.head == "cae":
        Q = 8                                              # 8 heads of the query token (cae_att.py:81)
        desc = desc.split(",")[0] + f", CAE attentive block (8 heads, LayerNorm-ed keys / values), {Cc} classes"
    if args.head == "siglip":
        Q = 8                                              # 8 heads of the latent query (attention_pool.py:24)
        desc = desc.split(",")[0] + f", SigLIP attention pool (8 heads, latent query, MLP x4), {Cc} classes"

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(D, Cc)
    torch.manual_seed(0)                                   # same init on every rank (DDP broadcasts rank 0's)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features=args.head, ep_queries=Q, d_out=1, nb_classes=Cc, num_heads=16,
                                                model="capi_vitl14_in1k" if Nn == 256 else "vit_base_patch16"))
    head = enc.head.to(dev).train()
    lr = 0.1 * (B * world) / 256                           # blr * eff_batch / 256 (main_linprobe.py:572-573)
    if args.arith != "fp32" and args.head not in ("ep", "coca", "abmilp"):
        raise SystemExit("--arith bf16_autocast is implemented for the EP, CoCa and AbMILP heads only")
    eng = make_engine(head, optimizer="lars", lr=lr, weight_decay=0.0, **({"arithmetic": args.arith} if args.arith != "fp32" else {}))
    if hasattr(eng, "defer_update"):
        eng.defer_update = True          # (takes effect only with EP_DEFER_OPT=1 -- measured slower, engine._can_defer; the loop
                                         # below calls flush() before it stops the clock, as train_one_epoch does)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    xs = [torch.randn(B, Nn, D, device=dev, generator=gen) for _ in range(args.buffers)]
    if args.tokens == "bf16":
        xs = [x.to(torch.bfloat16) for x in xs]
    esize = 2 if args.tokens == "bf16" else 4
    ts = [torch.randint(0, Cc, (B,), device=dev, generator=gen) for _ in range(args.buffers)]
    ts_all = ts

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Device spin-up, part of SETUP (untimed, reported as "spinup_steps"; --spinup 0 turns it off): from an idle chip the
    # step time is not monotone -- steps 3-5 run at the steady 0.45 ms, the power controller then pulls the clock down
    # (0.51-0.56 ms) and releases it over the next ~30 steps (profiles/r02/dvfs_transient.txt).  A training job lives in
    # the state behind that transient, so the W warm-up and K timed steps are taken there, not in its trough.
    for i in range(args.spinup):
        eng.train_step(xs[i % args.buffers], ts[i % args.buffers])
    for i in range(args.warmup):
        eng.train_step(xs[i % args.buffers], ts[i % args.buffers])
    eng.flush()                                             # (data parallel: the deferred half of a pipelined step)
    eng.read_stats()
    barrier()
    # spread of the step time from device events, no host sync.  An event is a marker packet the queue drains in front
    # of (a ~6 us hole per mark in the kernel timeline), so the marks sit every `mark_every` steps and a sample is the
    # mean step time of one such window
    me = max(1, args.mark_every)
    if args.steps // me < 5:                               # the spread comes from at least five windows
        me = max(1, args.steps // 5)
    nwin = args.steps // me
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(nwin + 1)]
    t0 = time.perf_counter()
    if nwin:
        marks[0].record()
    for i in range(args.steps):
        eng.train_step(xs[i % args.buffers], ts[i % args.buffers])
        if (i + 1) % me == 0 and (i + 1) // me <= nwin:
            marks[(i + 1) // me].record()
    eng.flush()                                             # all K steps complete inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    step_seq = [marks[i].elapsed_time(marks[i + 1]) / me for i in range(nwin)]
    if os.environ.get("EP_BENCH_DUMP_STEPS") and rank == 0:
        print("step_ms sequence:", " ".join("%.4f" % v for v in step_seq), file=sys.stderr)
    step_ms = sorted(step_seq) or [elapsed * 1e3 / args.steps]
    pct = lambda q: round(step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))], 4)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_sum, top1, _, bad = eng.read_stats()

    # ---- data parallel (N > 1): the one exchange step of the path alone, and the two schedules side by side ----
    dp_obj = None
    if world > 1 and args.head == "ep" and os.environ.get("EP_BENCH_DP_OBJECT", "1") != "0":
        # (a) the flat-gradient all-reduce of a step, event-bracketed in untimed steps (forward/backward, all-reduce, update
        # as three calls -- what train_step does for N > 1)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ar = []
        for i in range(12):
            eng.forward_backward(xs[i % args.buffers], ts[i % args.buffers])
            ea.record(); eng.all_reduce_grads(); eb.record()
            eng.optimizer_step()
            torch.cuda.synchronize()
            if i >= 2:
                ar.append(ea.elapsed_time(eb) * 1e3)
        eng.read_stats()
        ar.sort()
        ar_us = ar[len(ar) // 2]
        nbytes = int(eng.flat_g.numel()) * 4
        # (b) the overlapped schedule (engine.py: cls_token's bucket first, the large bucket beside the next first token pass;
        # opt-in) on a second engine, same protocol, same K -- `value` above stays the default schedule
        torch.manual_seed(0)
        enc_o = Enc()
        probe_heads.build_probe_head(enc_o, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=Cc, num_heads=16,
                                                      model="vit_base_patch16"))
        eng_o = make_engine(enc_o.head.to(dev).train(), optimizer="lars", lr=lr, weight_decay=0.0, overlap_comm=True)
        for i in range(args.spinup + args.warmup):
            eng_o.train_step(xs[i % args.buffers], ts[i % args.buffers])
        eng_o.flush(); eng_o.read_stats()
        barrier()
        to0 = time.perf_counter()
        for i in range(args.steps):
            eng_o.train_step(xs[i % args.buffers], ts[i % args.buffers])
        eng_o.flush()
        barrier()
        el_o = time.perf_counter() - to0
        t = torch.tensor([el_o], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_o = float(t.item())
        lo = eng_o.read_stats()[0]
        dp_obj = {"allreduce_us": round(ar_us, 2), "allreduce_bytes": nbytes,
                  "allreduce_busbw_GBs": round(2.0 * (world - 1) / world * nbytes / (ar_us * 1e-6) / 1e9, 2),
                  "allreduce_method": "torch.cuda events around the ONE sum all-reduce of the flat fp32 gradient buffer in 10 untimed steps (median)",
                  "overlap_comm": {"off": {"ms_per_step": round(elapsed / args.steps * 1e3, 4), "value": round(B * world * args.steps / elapsed, 1),
                                           "note": "the default schedule = the headline `value`"},
                                   "on": {"ms_per_step": round(el_o / args.steps * 1e3, 4), "value": round(B * world * args.steps / el_o, 1),
                                          "pipelined": bool(eng_o._pipelined), "mean_loss_over_timed_steps": round(lo / max(1, args.steps), 5)}}}
        del eng_o

    # ---- eval forward (reference engine_finetune.py:106-166 inner forward), outside the timed training region ----
    n_eval = max(10, args.steps // 2)
    eng.eval_logits(xs[0]); torch.cuda.synchronize()
    te0 = time.perf_counter()
    for i in range(n_eval):
        eng.eval_logits(xs[i % args.buffers])
    torch.cuda.synchronize()
    eval_s = (time.perf_counter() - te0) / n_eval

    # ---- dominant kernel: EP pooling passes, timed alone with HIP events on the launch stream ----
    def time_kernel(fn, iters):
        fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters

    imgq = args.head in ("simpool", "esimpool")            # per-image-query passes (csrc/ep_pool_imgq.hip)
    cbam = args.head == "cbam"                             # its own streaming passes (csrc/ep_cbam.hip)
    rowq = args.head == "clip"                             # full-width per-image query rows (ep_imgqf_kernel)
    if rowq:
        cls, scale = torch.randn(B, Q, D, device=dev) * 0.05, 1.0
        tstat = F_.token_stats(xs[0] if args.tokens == "f32" else xs[0].float(), 1e-6)
        sbias = torch.randn(B, Q, Nn, device=dev) * 0.1
    elif cbam:
        cls, scale = None, 1.0
    elif imgq:
        cls, scale = torch.randn(B, D, device=dev) * 0.05, 1.0
        tstat = F_.token_stats(xs[0] if args.tokens == "f32" else xs[0].float(), 1e-6)
    elif args.head in ("coca", "siglip", "cae", "jepa", "aim", "cait"):   # the same kernel, fed with the H derived query rows
        cls, scale = torch.randn(Q, D, device=dev) * 0.05, 1.0
    else:
        cls, scale = head[0].cls_token.detach(), head[0].scale
    keep = {}

    def run_fwd(i):
        if cbam:
            keep["out"] = F_.cbam_channel_table(xs[i % args.buffers])
            return
        if rowq:
            keep["out"] = F_.rowq_pool_forward(xs[i % args.buffers], cls, tstat, sbias)
        elif imgq:
            keep["out"] = F_.imgq_pool_forward(xs[i % args.buffers], cls, Q, tstat, args.head == "simpool")
        else:
            keep["out"] = F_.pool_forward(xs[i % args.buffers], cls, scale)
    t_fwd = time_kernel(run_fwd, args.kernel_iters)
    if cbam:
        P = torch.zeros(B, D, device=dev); S = ML = None
    elif imgq:
        P, ML = keep["out"]
        S = None
    else:
        P, S, ML = keep["out"]
        ML[:, :, 2] = 0.0
    dP = torch.randn_like(P)
    ws_bytes = eng.lib.ep_pool_workspace_bytes(B, Nn, D, Q)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    dcls = torch.empty(Q, D, device=dev)
    from efficient_probing_amd import _native as N_
    stream = N_.current_stream_ptr(dev)

    def run_bwd(i):
        x = xs[i % args.buffers]
        if cbam:
            F_.cbam_channel_table(x)
            return
        if rowq:
            F_.rowq_pool_backward(x, S, ML, dP, tstat, sbias, True)
            return
        if imgq:
            F_.imgq_pool_backward(x, cls, Q, P, ML, dP, tstat, args.head == "simpool")
            return
        N_.check(eng.lib.ep_pool_backward(x.data_ptr(), 1 if args.tokens == "bf16" else 0, Nn * D, 0, B, Nn, D, Q, float(scale), S.data_ptr(), ML.data_ptr(),
                                          dP.data_ptr(), dcls.data_ptr(), 0, ws.data_ptr(), ws_bytes, stream), "bwd")
    t_bwd = time_kernel(run_bwd, args.kernel_iters)

    # ---- the two token-pass launches INSIDE the step (they carry the in-pass contractions and the weight-gradient side
    # work): event-bracketed by the library in a few untimed, instrumented steps (ep_debug_set_pass_events)
    def pass_events(e_, toks_, ts_):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for e in evs:
            e.record()                                       # (creates the underlying hipEvent_t)
        torch.cuda.synchronize()
        fw, bw = [], []
        try:
            e_.lib.ep_debug_set_pass_events(*[e.cuda_event for e in evs])
            for i in range(12):
                e_.train_step(toks_[i % len(toks_)], ts_[i % len(ts_)])
                torch.cuda.synchronize()
                if i >= 2:
                    fw.append(evs[0].elapsed_time(evs[1]) * 1e3); bw.append(evs[2].elapsed_time(evs[3]) * 1e3)
        finally:
            e_.lib.ep_debug_set_pass_events(None, None, None, None)
        e_.flush(); e_.read_stats()
        fw.sort(); bw.sort()
        return {"fwd_us": round(fw[len(fw) // 2], 2), "bwd_us": round(bw[len(bw) // 2], 2), "samples": len(fw),
                "method": "hipEvents recorded by the library around the two pass launches of 10 untimed steps (median); "
                          "an event drains the queue, so each figure includes the ~2 us fill of an empty chip"}

    in_step = pass_events(eng, xs, ts) if args.head == "ep" else None

    algo_bytes = B * Nn * D * esize                           # one streaming read of the stored tokens
    dt = 1 if args.tokens == "bf16" else 0
    kname_f = "ep_imgqf_kernel (forward)" if rowq else "ep_cbam_chan_kernel (pass A)" if cbam else "ep_imgq_kernel (forward)" if imgq else eng.lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 0, dt).decode()
    kname_b = "ep_imgqf_kernel (backward)" if rowq else "ep_cbam_chan_kernel (pass A)" if cbam else "ep_imgq_kernel (backward)" if imgq else eng.lib.ep_pool_kernel_name_ex(B, Nn, D, Q, 1, dt).decode()
    # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of the
    # committed profile of this same command -- rocprofv3 counters cannot be read from inside the run
    traffic, traffic_source, traffic_stale = None, None, False
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):           # (the newest round that has one: it is current or it is stale)
        tpath = os.path.join(ROOT, "profiles", rnd, f"{args.workload}_hbm_traffic_pmc.json")
        if traffic is None and not traffic_stale and os.path.exists(tpath) and B == 1024 and args.tokens == "f32" and args.head == "ep":
            try:
                prof = json.load(open(tpath))
                pk = prof["per_kernel"]
                # Counters are a citation of a committed profile, never of this run -- and only of THESE kernel sources: the
                # profile carries the sha256 of csrc/ + include/ it was taken on (tools/src_hash.py, written by
                # tools/make_profiles.sh); any other tree makes the number stale and it is withheld.
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from src_hash import source_hash
                if prof.get("source_hash") != source_hash(ROOT):
                    traffic_stale = True
                    traffic_source = (f"profiles/{rnd}/{args.workload}_hbm_traffic_pmc.json was taken on other kernel sources "
                                      f"(source_hash {str(prof.get('source_hash'))[:12]} != {source_hash(ROOT)[:12]}): withheld")
                    break
                # of the dominant kernel (the second pass), its launches INSIDE train steps only where the profile separates them
                ent = next((v for k, v in pk.items() if k.startswith(kname_b)), None)
                if ent is not None:
                    traffic = round(ent.get("in_step", ent)["hbm_bytes_per_launch"])
                    traffic_source = (f"profiles/{rnd}/{args.workload}_hbm_traffic_pmc.json (committed rocprofv3 --pmc passes of this command on "
                                      f"these sources{', in-step launches only' if 'in_step' in ent else ''}; not re-measured in this run)")
            except Exception:
                traffic = None
    fwd_gbs = algo_bytes / t_fwd / 1e9
    bwd_gbs = algo_bytes / t_bwd / 1e9

    # ---- secondary runs of the SAME step (never `value`): a fresh head + engine per run, the same timing protocol
    def secondary(sN, sD, sQ, storage, steps, toks=None, sC=None, sB=None, passes=False, arithmetic="fp32"):
        sC = sC or Cc
        sB = sB or B
        torch.manual_seed(0)

        class Enc2(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.head = torch.nn.Linear(sD, sC)
        enc2 = Enc2()
        probe_heads.build_probe_head(enc2, Namespace(cls_features="ep", ep_queries=sQ, d_out=1, nb_classes=sC, num_heads=16,
                                                     model="vit_base_patch16"))
        ts = [(t % sC)[:sB].contiguous() for t in ts_all]
        h2 = enc2.head.to(dev).train()
        eng2 = make_engine(h2, optimizer="lars", lr=0.1 * (sB * world) / 256, weight_decay=0.0,
                           **({"arithmetic": arithmetic} if arithmetic != "fp32" else {}))
        if hasattr(eng2, "defer_update"):
            eng2.defer_update = True
        if toks is None:
            g2 = torch.Generator(device=dev).manual_seed(4321 + rank)
            toks = [torch.randn(sB, sN, sD, device=dev, generator=g2) for _ in range(args.buffers)]
        elif sB != B:
            toks = [x[:sB] for x in toks]                     # (a batch-strided view of the first sB images)
        if storage == "bf16":
            toks = [x.to(torch.bfloat16) for x in toks]
        es = 2 if storage == "bf16" else 4
        for i in range(args.spinup + (min(10, args.warmup) or 1)):      # (spin-up as in the headline run, then warm-up)
            eng2.train_step(toks[i % args.buffers], ts[i % args.buffers])
        eng2.flush(); eng2.read_stats()
        barrier()
        tb0 = time.perf_counter()
        for i in range(steps):
            eng2.train_step(toks[i % args.buffers], ts[i % args.buffers])
        eng2.flush()
        barrier()
        el2 = time.perf_counter() - tb0
        if world > 1:
            t2 = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            el2 = float(t2.item())
        l2 = eng2.read_stats()[0]
        c2_, sc2 = h2[0].cls_token.detach(), h2[0].scale
        tf2 = time_kernel(lambda i: F_.pool_forward(toks[i % args.buffers], c2_, sc2), args.kernel_iters)
        v2 = sB * world * steps / el2
        line = {"value": round(v2, 1), "unit": "images/s", "n_gpus": world, "steps": steps, "ms_per_step": round(el2 / steps * 1e3, 4),
                "tokens": sN, "dim": sD, "queries": sQ, "batch_per_gpu": sB, "token_storage": storage, "arithmetic": "f32" if arithmetic == "fp32" else arithmetic,
                "kernel": eng.lib.ep_pool_kernel_name_ex(sB, sN, sD, sQ, 0, 1 if storage == "bf16" else 0).decode(),
                "us_per_launch": round(tf2 * 1e6, 2), "algorithmic_bytes": sB * sN * sD * es,
                "frac": round(sB * sN * sD * es / tf2 / 1e9 / HBM_PEAK_GBS, 4),
                "step_frac": round(v2 / world * 2 * sN * sD * es / 1e9 / HBM_PEAK_GBS, 4),
                "mean_loss_over_timed_steps": round(l2 / steps, 5)}
        if passes:                                            # the two pass launches INSIDE this engine's step
            ins = pass_events(eng2, toks, ts)
            ab = sB * sN * sD * es
            line["in_step"] = {"fwd_us": ins["fwd_us"], "bwd_us": ins["bwd_us"],
                               "fwd_frac": round(ab / (ins["fwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "bwd_frac": round(ab / (ins["bwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "bwd_kernel": eng.lib.ep_pool_kernel_name_ex(sB, sN, sD, sQ, 1, 1 if storage == "bf16" else 0).decode()}
        return line

    default_ep = args.head == "ep" and args.tokens == "f32" and args.workload == "c2" and args.arith == "fp32"
    # (N = 1 only) the same tokens STORED as bf16 (fp32 arithmetic and results; the token passes run on the bf16
    # matrix cores, csrc/ep_pool_mb.hip)
    bf16_line = None
    if world == 1 and args.head == "ep" and args.tokens == "f32" and args.arith == "fp32" and not args.no_bf16_secondary:
        bf16_line = secondary(Nn, D, Q, "bf16", max(10, min(args.steps, 50)), toks=xs, passes=True)
    # the north-star shape (ViT-B/16 tokens 197x768, BASELINE.json north_star) beside the configs[1] headline
    ns_line = None
    if default_ep and not args.no_north_star:
        nsN, nsD, nsQ = WORKLOADS["ns"][:3]
        ns_line = secondary(nsN, nsD, nsQ, "f32", args.steps)
        ns_line["workload"] = WORKLOADS["ns"][4]

    # every other BASELINE configuration as a short secondary of the same step (never `value`)
    configs = None
    if default_ep and not args.no_configs:
        configs = {}
        for name in ("c1", "c3", "c4", "c5"):
            cN, cD, cQ, cC, cdesc = WORKLOADS[name]
            torch.cuda.empty_cache()
            line = secondary(cN, cD, cQ, "f32", 20, sC=cC, passes=(name == "c5"))
            line["workload"] = cdesc
            line["classes"] = cC
            if name == "c5":
                # a co-bound step: its six dense contractions (y, dP, dWv: 2 B D D each; logits, dz, dWc: 2 B D C each) run on the
                # bf16 matrix cores as six bf16 products per fp32 product (csrc/ep_planes.hip, ep_wgrad3.h); their time is the step
                # minus the two token passes (an upper bound: BatchNorm / CE / optimizer launches are in it too)
                flop = 3 * 2.0 * B * cD * cD + 3 * 2.0 * B * cD * cC
                t_us = line["ms_per_step"] * 1e3 - line["in_step"]["fwd_us"] - line["in_step"]["bwd_us"]
                line["mfma"] = {"kernel": "ep_gemm_planes_kernel + ep_gemm_b3_kernel", "flop_per_step": flop, "time_us": round(t_us, 1),
                                "achieved": round(flop / t_us / 1e6, 1), "peak": round(2500.0 / 6.0, 1), "unit": "TFLOP/s",
                                "frac": round(flop / t_us / 1e6 / (2500.0 / 6.0), 4),
                                "peak_note": "fp32-accurate product = six bf16 products: 2.5 PFLOP/s / 6; time = step - token passes"}
            configs[name] = line
        # the headline shape at the per-GPU batches a fixed global batch gives on more GPUs (the published protocol's effective
        # batch is 4096 -- reference README.md:119-120 -- i.e. 512 per GPU on 8): what a --global-batch point is read against
        for nb in (512, 256):
            if nb < B:
                line = secondary(Nn, D, Q, "f32", 20, toks=xs, sB=nb)
                line["workload"] = desc + f", batch {nb} per GPU"
                configs[f"{args.workload}_b{nb}"] = line
        # the published protocol's query count (--ep_queries 32, reference README.md:133-134: every published accuracy row) at
        # the headline shape, f32 and bf16-stored tokens -- BASELINE.json's configurations say q = 8, these say what the README's
        # own commands would run at (beyond D = 768 and for bf16 tokens the passes run in chunks of 16 queries)
        for stor in ("f32", "bf16"):
            line = secondary(Nn, D, 32, stor, 20, toks=xs, passes=True)
            line["workload"] = desc.replace("EP q=8", "EP q=32") + (" [tokens stored as bf16]" if stor == "bf16" else "")
            # co-bound passes at 32 queries: priced on the matrix axis as well (2 contractions x 2 B N D Q FLOP per pass; fp32
            # tokens on v_mfma_f32_16x16x4_f32: 157.3 TFLOP/s; bf16 tokens as three bf16 products per fp32 product: 2500 / 3)
            flop = 4.0 * B * Nn * D * 32
            peak = F32_MFMA_PEAK_TFLOPS if stor == "f32" else 2500.0 / 3.0
            ins = line["in_step"]
            line["mfma"] = {"kernel": line["kernel"], "flop_per_pass": flop, "peak": round(peak, 1), "unit": "TFLOP/s",
                            "fwd_achieved": round(flop / ins["fwd_us"] / 1e6, 1), "bwd_achieved": round(flop / ins["bwd_us"] / 1e6, 1),
                            "fwd_frac": round(flop / ins["fwd_us"] / 1e6 / peak, 4), "bwd_frac": round(flop / ins["bwd_us"] / 1e6 / peak, 4),
                            "peak_note": ("exact fp32 matrix instruction" if stor == "f32" else
                                          "bf16 tokens x fp32 operand as three bf16 terms: 2.5 PFLOP/s / 3")}
            configs[f"{args.workload}_q32" + ("_bf16" if stor == "bf16" else "")] = line
        # configs[4] as it fits 8 x 288 GB: the pre-dumped ViT-7B tokens stored as bf16, both passes in the step
        torch.cuda.empty_cache()
        cN, cD, cQ, cC, cdesc = WORKLOADS["c5"]
        line = secondary(cN, cD, cQ, "bf16", 20, sC=cC, passes=True)
        line["workload"] = cdesc + " [tokens stored as bf16, fp32 arithmetic]"
        line["classes"] = cC
        configs["c5_bf16"] = line
        # ... and at the other published rows' shapes: MAE ViT-L/16, SigLIP2 SO400M (fp32 tokens) and the
        # pre-dumped ViT-7B tokens as they fit 8 x 288 GB (bf16-stored)
        for cname, stor in (("c3", "f32"), ("c4", "f32"), ("c5", "bf16")):
            try:
                qN, qD, _, qC, qdesc = WORKLOADS[cname]
                torch.cuda.empty_cache()
                line = secondary(qN, qD, 32, stor, 20, sC=qC, passes=True)
                line["workload"] = qdesc.replace("EP q=8", "EP q=32") + (" [tokens stored as bf16]" if stor == "bf16" else "")
                base = configs.get(cname if stor == "f32" else cname + "_bf16", {}).get("ms_per_step")
                if base:
                    line["ratio_to_q8_step"] = round(line["ms_per_step"] / base, 3)
                configs[f"{cname}_q32" + ("_bf16" if stor == "bf16" else "")] = line
            except Exception as e:
                configs[f"{cname}_q32"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        # The published runs train under --amp bfloat16 (reference README.md:639-645, engine_finetune.py:52-55).  The same steps in
        # the AMP-bf16 arithmetic mode (ep_head_step.arith = EP_ARITH_BF16_AUTOCAST: the six contractions as ONE bf16 matrix-core
        # product with fp32 accumulation; token passes, softmax, BatchNorm, loss and optimizer unchanged) -- secondary objects,
        # never `value`: a different arithmetic from the headline's fp32 (tests/test_gpu_amp_bf16.py pins its distance from the
        # reference's bf16-autocast head)
        amp = {}
        try:
            l1 = secondary(Nn, D, Q, "bf16", 20, toks=xs, passes=True, arithmetic="bf16_autocast")
            l1["workload"] = desc + " [tokens stored as bf16, AMP-bf16 contractions]"
            amp[args.workload + "_bf16"] = l1
            l0 = secondary(Nn, D, Q, "f32", 20, toks=xs, passes=True, arithmetic="bf16_autocast")
            l0["workload"] = desc + " [AMP-bf16 contractions]"
            amp[args.workload] = l0
            # the published rows the mode refused until round 5 (query slices that are no multiple of 32 columns): --ep_queries 32
            # at the headline shape (slice 24) and SigLIP2 SO400M (1152 / 8 = 144)
            lq = secondary(Nn, D, 32, "bf16", 20, toks=xs, passes=True, arithmetic="bf16_autocast")
            lq["workload"] = desc.replace("EP q=8", "EP q=32") + " [tokens stored as bf16, AMP-bf16 contractions]"
            amp[args.workload + "_q32_bf16"] = lq
            torch.cuda.empty_cache()
            c4N, c4D, c4Q, c4C, c4desc = WORKLOADS["c4"]
            for stor in ("f32", "bf16"):
                l4 = secondary(c4N, c4D, c4Q, stor, 20, sC=c4C, passes=True, arithmetic="bf16_autocast")
                l4["workload"] = c4desc + (" [tokens stored as bf16, AMP-bf16 contractions]" if stor == "bf16" else " [AMP-bf16 contractions]")
                l4["classes"] = c4C
                amp["c4" + ("_bf16" if stor == "bf16" else "")] = l4
            torch.cuda.empty_cache()
            for stor in ("f32", "bf16"):
                l5 = secondary(cN, cD, cQ, stor, 20, sC=cC, passes=True, arithmetic="bf16_autocast")
                l5["workload"] = cdesc + (" [tokens stored as bf16, AMP-bf16 contractions]" if stor == "bf16" else " [AMP-bf16 contractions]")
                l5["classes"] = cC
                flop = 3 * 2.0 * B * cD * cD + 3 * 2.0 * B * cD * cC
                t_us = l5["ms_per_step"] * 1e3 - l5["in_step"]["fwd_us"] - l5["in_step"]["bwd_us"]
                l5["mfma"] = {"kernel": "ep_gemm_planes_kernel<NB, 1> + ep_gemm_b3_kernel (one product)", "flop_per_step": flop, "time_us": round(t_us, 1),
                              "achieved": round(flop / t_us / 1e6, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flop / t_us / 1e6 / 2500.0, 4),
                              "peak_note": "one bf16 product per multiply-add: 2.5 PFLOP/s; time = step - token passes"}
                amp["c5" + ("_bf16" if stor == "bf16" else "")] = l5
        except Exception as e:                                  # a secondary never takes the headline line down with it
            amp["error"] = f"{type(e).__name__}: {e}"[:300]
        configs["amp_bf16"] = amp
        torch.cuda.empty_cache()
        # BASELINE configs[3] compares three heads on the SO400M tokens: the other two (CoCa pooler: HBM-bound like EP; AbMILP:
        # matrix-core-bound) as short child runs of this file, rank 0 at N = 1 only (their own engines and workspaces)
        if world == 1:
            import subprocess
            torch.cuda.empty_cache()
            for hname, harith in (("coca", "fp32"), ("abmilp", "fp32"), ("coca", "bf16_autocast"), ("abmilp", "bf16_autocast")):
                cmd = [sys.executable, os.path.abspath(__file__), "--head", hname, "--workload", "c4", "--steps", "10", "--warmup", "3",
                       "--spinup", "5", "--no-cpu-baseline", "--no-configs", "--no-north-star", "--no-bf16-secondary",
                       "--no-through-engine", "--kernel-iters", "3", "--arith", harith]
                dest = configs if harith == "fp32" else configs.setdefault("amp_bf16", {})
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    dest["c4_" + hname] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "n_gpus": 1, "steps": d["steps"],
                                              "ms_per_step": d["ms_per_step"], "batch_per_gpu": d["config"]["batch_per_gpu"],
                                              "workload": d["config"]["workload"], "bound": d["roofline"]["bound"],
                                              "step_frac": d["roofline"].get("step_frac"),
                                              "mean_loss_over_timed_steps": d["check"]["mean_loss_over_timed_steps"]}
                    if d["roofline"]["bound"] == "mfma":
                        r_ = d["roofline"]
                        step_tf = d["value"] * r_["step_flop_per_image"] / 1e12
                        dest["c4_" + hname]["mfma"] = {"kernel": r_["kernel"], "achieved_tflops": r_["achieved"], "peak": r_["peak"],
                                                          "frac": r_["frac"], "step_achieved_tflops": round(step_tf, 1),
                                                          "step_frac": r_.get("step_frac"), "peak_note": r_.get("peak_note")}
                except Exception as e:                          # a secondary never takes the headline line down with it
                    dest["c4_" + hname] = {"error": f"{type(e).__name__}: {e}"[:200]}
        torch.cuda.empty_cache()
    # the same workload through the reference's loop surface (engine_finetune.train_one_epoch, reference
    # engine_finetune.py:22-103): a resident token store, adjust_learning_rate every iteration, meters every 20
    through = None
    stream_head = args.head != "ep" and args.workload == "c2" and args.tokens == "f32"     # (the matrix heads return above)
    if (default_ep or stream_head) and world == 1 and not args.no_through_engine:
        import contextlib
        from efficient_probing_amd import engine_finetune as EF
        from efficient_probing_amd.token_store import ResidentTokenStore
        from efficient_probing_amd.util.lars import LARS
        torch.manual_seed(0)
        enc3 = Enc()
        probe_heads.build_probe_head(enc3, Namespace(cls_features=args.head, ep_queries=Q, d_out=1, nb_classes=Cc, num_heads=16,
                                                     model="capi_vitl14_in1k" if Nn == 256 else "vit_base_patch16"))
        enc3.to(dev)
        store = ResidentTokenStore.from_tensors(torch.cat(xs, 0), torch.cat(ts, 0))
        n_it = args.engine_steps

        class Epochs:                                        # `n_it` iterations: the store's epochs back to back
            def __len__(self):
                return n_it

            def __iter__(self):
                k, ep = 0, 0
                while k < n_it:
                    for bt in store.batches(B, epoch=ep):
                        if k == n_it:
                            return
                        k += 1
                        yield bt
                    ep += 1
        opt3 = LARS(enc3.head.parameters(), lr=lr, weight_decay=0.0)
        a3 = Namespace(lr=lr, min_lr=0.0, warmup_epochs=10, epochs=90, accum_iter=1, amp="none")
        crit = torch.nn.CrossEntropyLoss()
        with contextlib.redirect_stdout(sys.stderr):
            n_it = 60
            EF.train_one_epoch(enc3, crit, Epochs(), opt3, dev, 20, None, args=a3)      # spin-up + warm-up, untimed
            n_it = args.engine_steps
            barrier()
            tq0 = time.perf_counter()
            st3 = EF.train_one_epoch(enc3, crit, Epochs(), opt3, dev, 21, None, args=a3)
            barrier()
            el3 = time.perf_counter() - tq0
        through = {"value": round(B * n_it / el3, 1), "unit": "images/s", "steps": n_it, "ms_per_step": round(el3 / n_it * 1e3, 4),
                   "surface": "efficient_probing_amd.engine_finetune.train_one_epoch over ResidentTokenStore batches "
                              "(image_index into HBM-resident tokens), LARS, lr_sched.adjust_learning_rate per iteration, "
                              "meters read back every 20 iterations",
                   "mean_loss": round(float(st3.get("loss", float("nan"))), 5)}
        tabs = sorted(k for k, _ in store.__dict__.get("_tables", {}))
        if tabs:                                             # functions of the frozen tokens, computed once per store
            through["store_tables"] = tabs
        del store
        torch.cuda.empty_cache()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        if through is not None:
            through["vs_train_step"] = round(through["value"] / value, 4)
        # The roofline object prices the DOMINANT kernel of the step as it runs IN the step: the second token pass
        # (it carries the in-pass dP tasks in front of its stream and the weight-gradient side workgroups behind it), measured
        # by the library's own HIP events around that launch.  The first pass and both passes launched alone are sub-objects.
        alone = {"fwd": {"kernel": kname_f, "us_per_launch": round(t_fwd * 1e6, 2), "achieved": round(fwd_gbs, 1),
                         "frac": round(fwd_gbs / HBM_PEAK_GBS, 4)},
                 "bwd": {"kernel": kname_b, "us_per_launch": round(t_bwd * 1e6, 2), "achieved": round(bwd_gbs, 1),
                         "frac": round(bwd_gbs / HBM_PEAK_GBS, 4)},
                 "note": "the same kernels launched alone (no in-pass tasks, no side workgroups), HIP events over --kernel-iters launches"}
        step_frac = round(value / world * 2 * Nn * D * esize / 1e9 / HBM_PEAK_GBS, 4)
        if in_step is not None:
            dom_us, dom_name, dom_how = in_step["bwd_us"], kname_b, "in the step: " + in_step["method"]
            fwd_us = in_step["fwd_us"]
        else:                                                  # heads without the pass events: the slower pass, launched alone
            dom_us, dom_name = (t_bwd * 1e6, kname_b) if t_bwd >= t_fwd else (t_fwd * 1e6, kname_f)
            dom_how, fwd_us = alone["note"], None
        dom_gbs = algo_bytes / (dom_us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                    "us_per_launch": round(dom_us, 2), "algorithmic_bytes": algo_bytes, "measured": dom_how,
                    "step_frac": step_frac, "alone": alone}
        if fwd_us is not None:
            roofline["fwd_kernel"] = {"kernel": kname_f, "us_per_launch": fwd_us,
                                      "achieved": round(algo_bytes / (fwd_us * 1e-6) / 1e9, 1),
                                      "frac": round(algo_bytes / (fwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        out = {
            "metric": {"ep": "EP-head train images/sec", "coca": "CoCa-head train images/sec",
                       "siglip": "SigLIP-head train images/sec", "cae": "CAE-head train images/sec",
                       "jepa": "JEPA-head train images/sec", "aim": "AIM-head train images/sec",
                       "simpool": "SimPool-head train images/sec", "esimpool": "eSimPool-head train images/sec",
                       "cait": "CaiT-head train images/sec", "clip": "CLIP-head train images/sec",
                       "cbam": "CBAM-head train images/sec"}[args.head], "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, **({"collective_backend": "gloo (EP_BENCH_SHARE_DEVICE=1: all ranks on one device, a test mode)"} if share else {}),
            "steps": args.steps, "warmup": args.warmup, "spinup_steps": args.spinup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if args.arith == "fp32" else "bf16_autocast (contractions: one bf16 product, fp32 accumulation; token passes, softmax, BatchNorm, loss, optimizer f32)",
            "data": "synthetic",
            "step_ms_p10": pct(0.10), "step_ms_p50": pct(0.50), "step_ms_p90": pct(0.90), "step_ms_window": me,
            "config": {"workload": desc + ("" if args.tokens == "f32" else " [tokens stored as bf16, fp32 arithmetic]"),
                       "tokens": Nn, "dim": D, "queries": Q, "classes": Cc, "batch_per_gpu": B, "token_storage": args.tokens,
                       "global_batch": B * world, "optimizer": "lars", "token_buffers": args.buffers,
                       "parallelism": f"dp{world}"},
            "roofline": roofline,
            "check": {"mean_loss_over_timed_steps": round(loss_sum / max(1, args.steps), 5),
                      "nonfinite_rows": bad,
                      "step_ms_device": {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90)}},
            "eval_forward": {"value": round(B / eval_s, 1), "unit": "images/s per GPU", "ms_per_batch": round(eval_s * 1e3, 4)},
        }
        if in_step is not None:
            in_step["fwd_frac"] = round(algo_bytes / (in_step["fwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            in_step["bwd_frac"] = round(algo_bytes / (in_step["bwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            in_step["between_and_after_us"] = round(ms_per_step * 1e3 - in_step["fwd_us"] - in_step["bwd_us"], 2)
            out["roofline"]["in_step"] = in_step
        if dp_obj is not None:
            out["data_parallel"] = dp_obj
        if configs is not None:
            out["configs"] = configs
        if through is not None:
            out["train_one_epoch"] = through
        if bf16_line is not None:
            out["bf16_token_storage"] = bf16_line
        if ns_line is not None:
            out["north_star"] = ns_line
        if world == 1 and not args.no_cpu_baseline:
            from oracle import torch_port
            cb = max(8, min(128, B))
            # torch-CPU degrades badly when over-subscribed on a many-core host: probe a few thread
            # counts for ~2 s each and time the sample at the best one (that count is reported as `cores`)
            ncpu = os.cpu_count() or 8
            cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
            mk = None
            if args.head == "coca":
                from oracle import coca_oracle
                mk = lambda: coca_oracle.make_head(D, Cc)
            if args.head == "siglip":
                from oracle import siglip_oracle
                mk = lambda: siglip_oracle.make_head(D, Cc)
            if args.head == "cae":
                from oracle import cae_oracle
                mk = lambda: cae_oracle.make_head(D, Cc)
            if args.head == "jepa":
                from oracle import jepa_oracle
                mk = lambda: jepa_oracle.make_head(D, Cc)
            if args.head == "aim":
                from oracle import aim_oracle
                mk = lambda: aim_oracle.make_head(D, Cc)
            if args.head == "cbam":
                from oracle import cbam_oracle
                mk = lambda: cbam_oracle.make_head(D, Cc)
            if args.head == "clip":
                from oracle import clip_oracle
                mk = lambda: clip_oracle.make_head(D, Cc, Nn)
            if args.head == "cait":
                from oracle import cait_oracle
                mk = lambda: cait_oracle.make_head(D, Cc)
            if args.head in ("simpool", "esimpool"):
                from oracle import simpool_oracle
                mk = lambda: simpool_oracle.make_head(D, Cc, args.head == "simpool")
            probe = {c: torch_port.time_train_steps(cb, Nn, D, Q, Cc, budget_s=2.0, threads=c, min_steps=1,
                                                    make=mk)["value"] for c in cands}
            best = max(probe, key=probe.get)
            r = torch_port.time_train_steps(cb, Nn, D, Q, Cc, budget_s=args.cpu_seconds, threads=best, make=mk)
            out["cpu_baseline"] = {"value": round(r["value"], 1), "unit": "images/s", "cores": r["threads"],
                                   "kind": "port", "host_cpus": ncpu,
                                   "sample": f"{r['steps']} train steps of batch {r['batch']} ({r['seconds']:.1f} s) "
                                             f"of the same workload, torch-CPU op-for-op port of the reference step; "
                                             f"threads chosen from {probe}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
