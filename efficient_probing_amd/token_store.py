"""Pre-dumped token store and feeder.

The reference can dump frozen-backbone tokens only as one small ``.npz`` (``tokens (n,N,C) float32``,
``images``, ``names``; reference tools/dump_tokens.py:95-98).  Training a probe on a 7B-parameter
encoder (BASELINE config 5) needs the whole dataset's tokens on disk once and then a reader that
keeps the head kernels fed.  This module provides

  * a sharded on-disk format: ``meta.json`` + ``tokens-XXXXX.bin`` (raw row-major ``(n_i, N, D)`` fp32 or bf16)
    + ``labels-XXXXX.npy``; shards are independent, so ranks partition them with no communication.  bf16 storage
    halves disk / HBM footprint (the 1.28 M x 196 x 4096 tokens of a ViT-7B fit 8 GPUs only in bf16); the token
    passes widen bf16 to fp32 on the fly and compute in fp32;
  * ``TokenStoreWriter`` / ``read_reference_npz`` (imports the reference's ``.npz`` dump);
  * ``ResidentTokenStore``: a rank's shard loaded ONCE into HBM (288 GB per MI355X: 163 k images of
    256x768 fp32 tokens per GPU, the whole 1.28 M-image ImageNet train set on 8 GPUs) and batches
    drawn as an int32 index vector -- the pooling kernels read the selected images IN PLACE
    (``image_index`` of the C ABI), so an epoch never copies or re-uploads a token;
  * ``StreamingTokenLoader``: for stores larger than HBM, double-buffered pinned-host -> device
    copies on a side stream (PCIe-bound, ~80 k img/s at 256x768 fp32; see EXPERIMENTS.md section 3).
"""
from __future__ import annotations

import json
import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

META = "meta.json"


class TokenStoreWriter:
    def __init__(self, out_dir: str, num_tokens: int, dim: int, shard_images: int = 8192, dtype: str = "float32"):
        if dtype not in ("float32", "bfloat16"):
            raise NotImplementedError("token store dtype: float32 or bfloat16")
        if dtype == "bfloat16" and dim % 8 != 0:
            raise ValueError("bfloat16 token store: dim must be a multiple of 8")
        os.makedirs(out_dir, exist_ok=True)
        self.out_dir, self.N, self.D, self.shard_images, self.dtype = out_dir, num_tokens, dim, shard_images, dtype
        self.shards: List[dict] = []
        self._tok: List[np.ndarray] = []
        self._lab: List[np.ndarray] = []
        self._n = 0

    def add(self, tokens: np.ndarray, labels: np.ndarray) -> None:
        tokens = np.ascontiguousarray(tokens, dtype=np.float32)
        assert tokens.ndim == 3 and tokens.shape[1:] == (self.N, self.D), tokens.shape
        assert len(labels) == len(tokens)
        self._tok.append(tokens); self._lab.append(np.asarray(labels, dtype=np.int64)); self._n += len(tokens)
        while self._n >= self.shard_images:
            self._flush(self.shard_images)

    def _flush(self, n: int) -> None:
        tok = np.concatenate(self._tok); lab = np.concatenate(self._lab)
        idx = len(self.shards)
        _encode(tok[:n], self.dtype).tofile(os.path.join(self.out_dir, f"tokens-{idx:05d}.bin"))
        np.save(os.path.join(self.out_dir, f"labels-{idx:05d}.npy"), lab[:n])
        self.shards.append({"index": idx, "images": int(n)})
        self._tok, self._lab, self._n = ([tok[n:]] if len(tok) > n else []), ([lab[n:]] if len(lab) > n else []), len(tok) - n

    def close(self) -> dict:
        if self._n:
            self._flush(self._n)
        meta = {"format": "ep-token-store-v1", "num_tokens": self.N, "dim": self.D, "dtype": self.dtype,
                "shards": self.shards, "total_images": int(sum(s["images"] for s in self.shards))}
        with open(os.path.join(self.out_dir, META), "w") as f:
            json.dump(meta, f, indent=1)
        return meta


def _encode(tok: np.ndarray, dtype: str) -> np.ndarray:
    """fp32 -> stored element type; bf16 (round to nearest even) travels as its raw 16-bit pattern."""
    if dtype == "float32":
        return tok
    return torch.from_numpy(np.ascontiguousarray(tok)).to(torch.bfloat16).view(torch.int16).numpy()


def _to_torch(arr: np.ndarray, dtype: str) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(arr))
    return t.view(torch.bfloat16) if dtype == "bfloat16" else t


def read_reference_npz(path: str) -> Tuple[np.ndarray, List[str]]:
    """Tokens and names of a reference dump (tools/dump_tokens.py:95-98: keys tokens/images/names)."""
    z = np.load(path, allow_pickle=True)
    return np.asarray(z["tokens"], dtype=np.float32), [str(n) for n in z["names"]]


def load_meta(store_dir: str) -> dict:
    with open(os.path.join(store_dir, META)) as f:
        meta = json.load(f)
    if meta.get("format") != "ep-token-store-v1":
        raise ValueError(f"{store_dir}: not an ep token store")
    return meta


def shards_of_rank(meta: dict, world: int, rank: int) -> List[dict]:
    """Round-robin shard ownership: independent units, no data-path collective."""
    return [s for s in meta["shards"] if s["index"] % world == rank]


def steps_per_epoch(meta: dict, world: int, batch_size: int, per_shard: bool = False) -> int:
    """Batches EVERY rank yields per epoch: the minimum over the ranks of what each one's shards hold, computed from
    ``meta.json`` alone (no collective).  Whole shards are dealt round-robin, so ranks can own different image counts;
    every engine step ends in a blocking all-reduce of the flat gradients, so all ranks must run the same number of
    steps or the longer one hangs there.  The reference gets equal lengths from ``DistributedSampler`` (it pads by
    wrap-around, main_linprobe.py:286-287); here the surplus of the larger ranks is dropped, like ``drop_last``.
    ``per_shard``: batches never span shards (the streaming loader)."""
    counts = []
    for r in range(world):
        mine = shards_of_rank(meta, world, r)
        counts.append(sum(s["images"] // batch_size for s in mine) if per_shard
                      else sum(s["images"] for s in mine) // batch_size)
    return min(counts) if counts else 0


def open_shard(store_dir: str, meta: dict, shard: dict) -> Tuple[np.memmap, np.ndarray]:
    tok = np.memmap(os.path.join(store_dir, f"tokens-{shard['index']:05d}.bin"),
                    dtype=np.float32 if meta.get("dtype", "float32") == "float32" else np.int16, mode="r",
                    shape=(shard["images"], meta["num_tokens"], meta["dim"]))
    lab = np.load(os.path.join(store_dir, f"labels-{shard['index']:05d}.npy"))
    return tok, lab


class ResidentTokenStore:
    """This rank's shards, resident in HBM.  ``batches(batch_size, epoch)`` yields
    ``(store_tensor, image_index int32 (B,), targets int64 (B,))``: pass ``store_tensor`` as the tokens and
    ``image_index`` to ``ProbeHeadEngine.train_step`` -- no per-batch copy happens."""

    def __init__(self, store_dir: str, device, world: int = 1, rank: int = 0, seed: int = 0):
        meta = load_meta(store_dir)
        mine = shards_of_rank(meta, world, rank)
        n = sum(s["images"] for s in mine)
        self.meta, self.world = meta, world
        self.N, self.D, self.seed, self.rank = meta["num_tokens"], meta["dim"], seed, rank
        self.dtype = meta.get("dtype", "float32")
        tdt = torch.float32 if self.dtype == "float32" else torch.bfloat16
        self.tokens = torch.empty((max(n, 1), self.N, self.D), device=device, dtype=tdt)
        labels = []
        off = 0
        for s in mine:
            tok, lab = open_shard(store_dir, meta, s)
            self.tokens[off:off + s["images"]].copy_(_to_torch(tok, self.dtype))
            labels.append(lab); off += s["images"]
        self.num_images = n
        self.labels = torch.from_numpy(np.concatenate(labels) if labels else np.zeros(0, np.int64)).to(device)

    @classmethod
    def from_tensors(cls, tokens: torch.Tensor, labels: torch.Tensor, seed: int = 0) -> "ResidentTokenStore":
        """A resident store around tokens that are already in HBM (synthetic benchmarks, tokens produced by a live encoder
        pass): ``tokens`` (n, N, D) fp32 / bf16 on the GPU, ``labels`` (n,) int64.  One rank's view (world = 1)."""
        self = cls.__new__(cls)
        n, N, D = tokens.shape
        self.meta, self.world, self.N, self.D, self.seed, self.rank = {"num_tokens": N, "dim": D}, 1, N, D, seed, 0
        self.dtype = "float32" if tokens.dtype == torch.float32 else "bfloat16"
        self.tokens, self.num_images = tokens, n
        self.labels = labels.to(device=tokens.device, dtype=torch.int64)
        return self

    def table(self, kind: str, eps: Optional[float] = None) -> torch.Tensor:
        """A per-token / per-image table over the WHOLE store, computed once and kept (indexed like ``tokens``, so a batch
        read in place through ``image_index`` finds its rows): functions of the frozen tokens alone, which the token passes
        of some heads would otherwise recompute from a third read of the batch in every step --
        ``"token_stats"``: (n, N, 2) {mean, rstd} of a LayerNorm over D with ``eps`` (CAE / JEPA / CaiT / CLIP / SimPool heads;
        the reference applies that LayerNorm to the tokens in every forward, e.g. poolings/cae/cae_att.py:104,
        poolings/simpool.py:52), ``"channel_stats"``: (n, 2, D) per-image column mean and sum of squared deviations (AIM's token
        BatchNorm, SimPool's mean-token query), ``"cbam_channel_table"``: (n, 3, D), ``"xhat_mean"``: (n, D) mean normalised token
        row (the CLIP head's query input; made from ``"token_stats"`` of the same ``eps``).  ``engine.attach_store(store)`` makes an
        engine look them up (``engine_finetune.train_one_epoch`` / ``evaluate`` do that for a ``StoreBatch``)."""
        from . import functional as F_
        key = (kind, None if eps is None else float(eps))
        cache = self.__dict__.setdefault("_tables", {})
        if key not in cache:
            fn = {"token_stats": lambda t, lo: F_.token_stats(t, eps), "channel_stats": lambda t, lo: F_.channel_stats(t),
                  "cbam_channel_table": lambda t, lo: F_.cbam_channel_table(t),
                  "xhat_mean": lambda t, lo: F_.token_xhat_mean(t, self.table("token_stats", eps)[lo:lo + t.shape[0]])}.get(kind)
            if fn is None:
                raise ValueError(f"unknown store table {kind!r}")
            n = self.tokens.shape[0]
            parts = [fn(self.tokens[lo:lo + 32768], lo) for lo in range(0, n, 32768)]  # bounded launch grids
            cache[key] = parts[0] if len(parts) == 1 else torch.cat(parts)
        return cache[key]

    def loader(self, batch_size: int, epoch: int = 0, shuffle: bool = True, drop_last: bool = True) -> "StoreEpoch":
        """One epoch as a sized iterable of ``StoreBatch(store_tensor, image_index, targets)`` -- what
        ``engine_finetune.train_one_epoch`` / ``evaluate`` take in place of a DataLoader (they need ``len()``)."""
        return StoreEpoch(self, batch_size, epoch, shuffle, drop_last)

    def num_batches(self, batch_size: int, drop_last: bool = True) -> int:
        """Batches per epoch.  Data parallel (world > 1): the same number on every rank (``steps_per_epoch``)."""
        if self.world > 1:
            return steps_per_epoch(self.meta, self.world, batch_size)
        return self.num_images // batch_size if drop_last else -(-self.num_images // batch_size)

    def batches(self, batch_size: int, epoch: int = 0, shuffle: bool = True, drop_last: bool = True):
        g = torch.Generator(device="cpu").manual_seed(self.seed * 1000003 + epoch * 101 + self.rank)
        order = torch.randperm(self.num_images, generator=g) if shuffle else torch.arange(self.num_images)
        order = order.to(dtype=torch.int32)
        if self.tokens.is_cuda:
            # through pinned memory, asynchronously: a copy from pageable memory blocks the host until the queue has drained
            # (one step latency per epoch -- visible when a store holds only a few batches); the host copy is kept until the
            # next epoch's replaces it
            self._order_host = order.pin_memory()
            order = self._order_host.to(self.tokens.device, non_blocking=True)
        if self.world > 1:                 # equal step counts on all ranks: each step all-reduces the gradients
            stop = self.num_batches(batch_size) * batch_size
        else:
            stop = self.num_images - (self.num_images % batch_size if drop_last else 0)
        labels = self.labels[order.long()]          # ONE gather per epoch: a batch is then two views, no kernel of its own
        for lo in range(0, stop, batch_size):
            yield StoreBatch(self.tokens, order[lo:lo + batch_size], labels[lo:lo + batch_size], owner=self)


class StoreBatch(tuple):
    """``(store_tensor, image_index, targets)``: a batch of a resident token store, read in place through the int32
    ``image_index``.  A type of its own so that ``engine_finetune.train_one_epoch`` / ``evaluate`` can tell it from a loader
    that yields ``(images, extra, target)`` -- for any other 3-tuple they keep the reference's ``batch[0]`` / ``batch[-1]``
    reading (reference engine_finetune.py:125-126,185-186)."""
    def __new__(cls, store, image_index, targets, owner=None):
        self = super().__new__(cls, (store, image_index, targets))
        self.owner = owner          # the ResidentTokenStore (its cached per-token / per-image tables: ``table()``), or None
        return self


class StoreEpoch:
    """``len()`` + iteration over one epoch of a ResidentTokenStore (see ``ResidentTokenStore.loader``)."""

    def __init__(self, store: ResidentTokenStore, batch_size: int, epoch: int, shuffle: bool, drop_last: bool):
        self.store, self.batch_size, self.epoch, self.shuffle, self.drop_last = store, batch_size, epoch, shuffle, drop_last

    def __len__(self) -> int:
        return self.store.num_batches(self.batch_size, self.drop_last)

    def __iter__(self):
        return self.store.batches(self.batch_size, self.epoch, self.shuffle, self.drop_last)


class StreamingTokenLoader:
    """Shards larger than HBM: memmap -> pinned staging -> device on a copy stream, two batches deep."""

    def __init__(self, store_dir: str, device, batch_size: int, world: int = 1, rank: int = 0):
        self.dir, self.device, self.B = store_dir, device, batch_size
        self.meta = load_meta(store_dir)
        self.shards = shards_of_rank(self.meta, world, rank)
        # data parallel: every rank stops after the same number of batches (see steps_per_epoch)
        self.limit = steps_per_epoch(self.meta, world, batch_size, per_shard=True) if world > 1 else None
        self.stream = torch.cuda.Stream(device=device)
        N, D = self.meta["num_tokens"], self.meta["dim"]
        self.dtype = self.meta.get("dtype", "float32")
        tdt = torch.float32 if self.dtype == "float32" else torch.bfloat16
        self.stage = [torch.empty((batch_size, N, D), dtype=tdt).pin_memory() for _ in range(2)]
        self.dev = [torch.empty((batch_size, N, D), dtype=tdt, device=device) for _ in range(2)]

    def __len__(self):
        n = sum(s["images"] // self.B for s in self.shards)
        return n if self.limit is None else min(n, self.limit)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        k = 0
        pending = None
        copied = [None, None]     # H2D-complete events per buffer (host staging may be rewritten after them)
        consumed = [None, None]   # main-stream events: the consumer's kernels on a device buffer are enqueued before them
        main = torch.cuda.current_stream(self.device)
        total = len(self)
        for s in self.shards:
            if k >= total:
                break
            tok, lab = open_shard(self.dir, self.meta, s)
            for lo in range(0, s["images"] - self.B + 1, self.B):
                if k >= total:
                    break
                buf = k & 1
                if copied[buf] is not None:
                    copied[buf].synchronize()                     # previous H2D out of this staging buffer is done
                self.stage[buf].copy_(_to_torch(tok[lo:lo + self.B], self.dtype))
                with torch.cuda.stream(self.stream):
                    if consumed[buf] is not None:
                        self.stream.wait_event(consumed[buf])     # consumer finished with this device buffer
                    self.dev[buf].copy_(self.stage[buf], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(self.stream)
                copied[buf] = ev
                cur = (buf, torch.from_numpy(lab[lo:lo + self.B]).to(self.device), ev)
                if pending is not None:
                    main.wait_event(pending[2])
                    yield self.dev[pending[0]], pending[1]
                    done = torch.cuda.Event(); done.record(main); consumed[pending[0]] = done
                pending = cur
                k += 1
        if pending is not None:
            main.wait_event(pending[2])
            yield self.dev[pending[0]], pending[1]
