"""Data-parallel plumbing of the probe head: one process per GPU, ONE all-reduce per step.

The reference wraps the model in DistributedDataParallel (reference main_linprobe.py:581-583):
every backward all-reduces (averages) the head gradients in buckets, and every forward broadcasts
the BatchNorm buffers from rank 0.  Here the head's parameters live in one flat fp32 buffer with a
gradient buffer of the same layout, so a step needs exactly one ``all_reduce(SUM)`` over that flat
buffer (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests); the division by the
world size is folded into the optimizer kernel's ``inv_scale``.  BatchNorm statistics stay local to
a rank (no SyncBN in the reference either) and running stats are synchronised from rank 0 only at
evaluation / checkpoint time.

Nothing in this module touches the GPU, so it is exercised on CPU with world_size 2 (gloo).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _native as N


def head_param_layout(D: int, Q: int, d_out: int, num_classes: int) -> Tuple[List[int], int]:
    """Offsets (in elements) of cls_token | v.weight | fc.weight | fc.bias in the flat buffer and the
    total length, as defined by the C ABI (``ep_head_param_offsets``): nn.Module.parameters()
    order, every tensor starting at a multiple of 4 elements."""
    lib = N.load()
    dims = N.EPHeadDims(B=1, N=1, D=D, Q=Q, d_out=d_out, C=num_classes)
    offs = (C.c_int64 * 4)()
    total = int(lib.ep_head_param_offsets(C.byref(dims), offs))
    return list(offs), total


def head_param_shapes(D: int, Q: int, d_out: int, num_classes: int):
    Dp = D // d_out
    return [(1, Q, D), (Dp, D), (num_classes, Dp), (num_classes,)]


def pack_flat(tensors: Sequence[torch.Tensor], offsets: Sequence[int], total: int, out: Optional[torch.Tensor] = None):
    """Copy per-parameter tensors into the flat layout (padding stays zero)."""
    if out is None:
        out = torch.zeros(total, dtype=torch.float32, device=tensors[0].device)
    for t, o in zip(tensors, offsets):
        out[o:o + t.numel()].copy_(t.reshape(-1))
    return out


def unpack_flat(flat: torch.Tensor, offsets: Sequence[int], shapes) -> List[torch.Tensor]:
    out = []
    for o, shp in zip(offsets, shapes):
        n = 1
        for s in shp:
            n *= s
        out.append(flat[o:o + n].view(shp))
    return out


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def all_reduce_flat_grads(flat_grads: torch.Tensor, group=None) -> float:
    """THE collective of a data-parallel step: in-place SUM all-reduce of the flat gradient buffer.
    Returns the factor the optimizer must apply to turn the sum into DDP's average (1 / world)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w


def broadcast_from_rank0(tensors: Sequence[torch.Tensor], group=None) -> None:
    """What DDP does at wrap time (parameters) and what the reference does every forward for the
    BatchNorm buffers; we call it once at start-up and at eval / checkpoint time."""
    if world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=0, group=group)


def shard_range(n_items: int, world: int, rk: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for rank rk: every rank gets the same count (the last
    ones are padded by wrap-around in DistributedSampler; here the remainder is dropped like
    ``drop_last`` so that per-rank BatchNorm sees equal batches)."""
    per = n_items // world
    return rk * per, (rk + 1) * per
