"""Data-parallel path on CPU: two processes, gloo backend.

What the GPU engine does per step is (local fwd/bwd into a flat gradient buffer) -> ONE sum
all-reduce of that buffer -> optimizer with inv_scale = 1/world.  The same plumbing
(efficient_probing_amd.parallel: flat layout from the C ABI, all_reduce_flat_grads,
broadcast_from_rank0) is driven here with the numpy oracle standing in for the kernels, and the
result is checked against the reference's DDP semantics evaluated in one process: per-rank
BatchNorm statistics, gradients averaged over ranks, LARS on the averaged gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case():
    from cases import Case
    return Case("dp", B=8, N=12, D=64, Q=4, C=10, seed=3, weight_decay=1e-4)


def _local_grads(case, inp, lo, hi):
    from oracle import ep_oracle as O
    st = O.HeadState(cls_token=inp["cls_token"].copy(), v_weight=inp["v_weight"].copy(),
                     fc_weight=inp["fc_weight"].copy(), fc_bias=inp["fc_bias"].copy(),
                     running_mean=np.zeros(case.D, np.float32), running_var=np.ones(case.D, np.float32),
                     num_queries=case.Q, d_out=1)
    out, cache = O.head_forward_train(st, inp["x_buf"][lo:hi], inp["targets"][lo:hi])
    g = O.head_backward(st, cache)
    return st, [g[k] for k in O.PARAM_ORDER], cache["new_bn"]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cases import make_inputs
        from oracle import ep_oracle as O
        from efficient_probing_amd import parallel as PAR
        case = _case()
        inp = make_inputs(case)
        offs, total = PAR.head_param_layout(case.D, case.Q, 1, case.C)
        shapes = PAR.head_param_shapes(case.D, case.Q, 1, case.C)
        # rank 1 starts from perturbed parameters: the start-up broadcast must overwrite them
        params = [torch.from_numpy(inp[k].copy()) for k in ("cls_token", "v_weight", "fc_weight", "fc_bias")]
        flat_p = PAR.pack_flat(params, offs, total)
        if rank != 0:
            flat_p += 1.0
        PAR.broadcast_from_rank0([flat_p])
        got = PAR.unpack_flat(flat_p, offs, shapes)
        for t, k in zip(got, ("cls_token", "v_weight", "fc_weight", "fc_bias")):
            assert np.array_equal(t.numpy(), inp[k]), k
        lo, hi = PAR.shard_range(case.B, world, rank)
        st, grads, _ = _local_grads(case, inp, lo, hi)
        flat_g = PAR.pack_flat([torch.from_numpy(g.copy()) for g in grads], offs, total)
        inv = PAR.all_reduce_flat_grads(flat_g)                      # ONE collective
        assert inv == 1.0 / world
        avg = [t.numpy() * np.float32(inv) for t in PAR.unpack_flat(flat_g, offs, shapes)]
        ps, mus = O.lars_step(st.params(), avg, [None] * 4, lr=0.5, weight_decay=case.weight_decay)
        q.put((rank, [p.copy() for p in ps]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_step_matches_ddp_semantics():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # single-process evaluation of the reference's DDP semantics
    from cases import make_inputs
    from oracle import ep_oracle as O
    case = _case()
    inp = make_inputs(case)
    per_rank = [_local_grads(case, inp, *(r * case.B // world, (r + 1) * case.B // world)) for r in range(world)]
    avg = [sum(g[i] for _, g, _ in per_rank) / np.float32(world) for i in range(4)]
    ps, _ = O.lars_step(per_rank[0][0].params(), avg, [None] * 4, lr=0.5, weight_decay=case.weight_decay)
    for r in range(world):
        for got, want in zip(results[r], ps):
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
    # both ranks hold identical parameters after the step (replicas stay in sync)
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
