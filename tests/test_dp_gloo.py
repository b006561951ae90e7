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


# ---- four ranks, an uneven last shard: per-rank resident-store loaders + shard_range + the flat all-reduce ------------------
def _store_case():
    from cases import Case
    return Case("dp4", B=6, N=10, D=32, Q=4, C=7, seed=9, weight_decay=0.0)


def _worker4(rank, world, port, q, store_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cases import make_inputs
        from oracle import ep_oracle as O
        from efficient_probing_amd import parallel as PAR, token_store as TS
        case = _store_case()
        inp = make_inputs(case)
        offs, total = PAR.head_param_layout(case.D, case.Q, 1, case.C)
        shapes = PAR.head_param_shapes(case.D, case.Q, 1, case.C)
        store = TS.ResidentTokenStore(store_dir, "cpu", world=world, rank=rank, seed=4)
        meta = TS.load_meta(store_dir)
        n_steps = TS.steps_per_epoch(meta, world, case.B)
        st = O.HeadState(cls_token=inp["cls_token"].copy(), v_weight=inp["v_weight"].copy(), fc_weight=inp["fc_weight"].copy(),
                         fc_bias=inp["fc_bias"].copy(), running_mean=np.zeros(case.D, np.float32),
                         running_var=np.ones(case.D, np.float32), num_queries=case.Q, d_out=1)
        seen = []
        steps = 0
        for tokens, idx, tgt in store.batches(case.B, epoch=0):
            x = tokens[idx.long()].numpy()
            seen.append((idx.numpy().copy(), tgt.numpy().copy()))
            out, cache = O.head_forward_train(st, x, tgt.numpy())
            g = O.head_backward(st, cache)
            flat_g = PAR.pack_flat([torch.from_numpy(g[k].copy()) for k in O.PARAM_ORDER], offs, total)
            inv = PAR.all_reduce_flat_grads(flat_g)                  # ONE collective per step: hangs if a rank ran fewer steps
            avg = [t.numpy() * np.float32(inv) for t in PAR.unpack_flat(flat_g, offs, shapes)]
            ps, _ = O.lars_step(st.params(), avg, [None] * 4, lr=0.3, weight_decay=0.0)
            st.cls_token, st.v_weight, st.fc_weight, st.fc_bias = ps
            steps += 1
        assert steps == n_steps, (steps, n_steps)
        q.put((rank, store.num_images, steps, [p.copy() for p in st.params()],
               np.concatenate([s[0] for s in seen]), store.labels.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_four_ranks_with_an_uneven_last_shard(tmp_path):
    """Whole shards are dealt round-robin, so four ranks over five shards (the last one short) own 2 / 1 / 1 / 1 shards and
    different image counts: every rank must run steps_per_epoch(meta, world, batch) steps -- the minimum over the ranks, from
    meta.json alone -- or the flat-gradient all-reduce of the longer ranks would hang; replicas stay identical; shard_range
    splits a global batch evenly."""
    from efficient_probing_amd import parallel as PAR, token_store as TS
    case = _store_case()
    rng = np.random.default_rng(2)
    w = TS.TokenStoreWriter(str(tmp_path), case.N, case.D, shard_images=16)
    tok = rng.standard_normal((16 * 4 + 9, case.N, case.D), dtype=np.float32)          # five shards: 16 16 16 16 9
    lab = rng.integers(0, case.C, tok.shape[0])
    w.add(tok, lab); w.close()
    meta = TS.load_meta(str(tmp_path))
    world = 4
    owned = [sum(s["images"] for s in TS.shards_of_rank(meta, world, r)) for r in range(world)]
    assert sorted(owned) == [16, 16, 16, 25] and sum(owned) == tok.shape[0]
    assert TS.steps_per_epoch(meta, world, case.B) == 16 // case.B == 2
    # shard_range: equal contiguous pieces, remainder dropped
    pieces = [PAR.shard_range(26, world, r) for r in range(world)]
    assert pieces == [(0, 6), (6, 12), (12, 18), (18, 24)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert [g[1] for g in got] == owned and all(g[2] == 2 for g in got)
    for g in got[1:]:                                           # replicas in sync after two steps
        for a, b in zip(got[0][3], g[3]):
            assert np.array_equal(a, b)
    for g in got:                                               # a rank draws each of its images at most once per epoch
        assert len(np.unique(g[4])) == len(g[4]) == 2 * case.B and g[4].max() < g[1]
