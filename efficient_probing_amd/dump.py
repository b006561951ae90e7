"""Token dump: drive a frozen encoder over a data loader into a sharded token store.

Counterpart of the reference's ``tools/dump_tokens.py`` (:60-98 there: 35 images, one ``.npz``, loaders
timm / openclip / hub_dinov2 / vit), sized for the protocol instead of for a figure: any ``token_fn(images) -> (B, N, D)``
(the frozen backbone on the stock PyTorch-ROCm path, SURVEY.md section 8: token extraction is not a native kernel) and any
iterable of ``(images, ..., targets)`` batches go through ``token_store.TokenStoreWriter`` into ``meta.json`` +
``tokens-XXXXX.bin`` + ``labels-XXXXX.npy`` shards, fp32 or bf16 -- what ``ResidentTokenStore`` / ``StreamingTokenLoader``
then feed to the probe-head kernels (BASELINE configs[4]: ViT-7B tokens pre-dumped, bf16 so that they fit 8 x 288 GB).

    python -m efficient_probing_amd.dump --out DIR --dtype bf16 --encoder pkg.mod:make_token_fn --data pkg.mod:make_loader
    python -m efficient_probing_amd.dump --out DIR --from-npz dump.npz [--labels labels.npy]     # a reference dump
    python -m efficient_probing_amd.dump --out DIR --synthetic 4096 --tokens 196 --dim 768      # plumbing / benchmarks

``--encoder`` names a callable ``make_token_fn(device) -> token_fn`` and ``--data`` a callable ``make_loader() -> iterable``
(``module:attribute``); the backbones and datasets themselves are out of scope here (timm / torchvision / open_clip are not
part of this package), which is why they are plugged in by name rather than enumerated as in the reference's ``--loader``.
"""
from __future__ import annotations

import argparse
import importlib
import sys
from typing import Callable, Iterable, Optional

import numpy as np
import torch

from .token_store import TokenStoreWriter, read_reference_npz

DTYPES = {"f32": "float32", "fp32": "float32", "float32": "float32", "bf16": "bfloat16", "bfloat16": "bfloat16"}


def token_view(tokens: torch.Tensor) -> torch.Tensor:
    """What an encoder hands back -> (B, N, D): (B, H, W, C) / (B, C, H, W) feature maps are flattened to token rows the
    way reference tools/dump_tokens.py:51-53 does it; (B, N, D) passes through."""
    if tokens.dim() == 4:
        b, a, c, d = tokens.shape
        tokens = tokens.flatten(2).transpose(1, 2) if (a > d and c == d) else tokens.reshape(b, a * c, d)
    if tokens.dim() != 3:
        raise ValueError(f"token_fn must return (B, N, D) tokens (or a 4-D feature map), got {tuple(tokens.shape)}")
    return tokens


@torch.no_grad()
def dump_tokens(data_loader: Iterable, token_fn: Callable[[torch.Tensor], torch.Tensor], out_dir: str, *,
                dtype: str = "bfloat16", shard_images: int = 8192, device: Optional[torch.device] = None,
                max_images: Optional[int] = None, log_every: int = 0) -> dict:
    """Run ``token_fn`` over every batch of ``data_loader`` (``batch[0]`` the images, ``batch[-1]`` the targets, as the
    reference's loops read a batch: engine_finetune.py:185-186) and write the tokens + labels as a sharded store.
    Returns the store's ``meta`` dict.  The writer is created on the first batch (it fixes N and D)."""
    dtype = DTYPES.get(dtype, dtype)
    writer, seen = None, 0
    for it, batch in enumerate(data_loader):
        images, target = batch[0], batch[-1]
        if device is not None:
            images = images.to(device, non_blocking=True)
        tok = token_view(token_fn(images)).float()
        if max_images is not None and seen + tok.shape[0] > max_images:
            tok, target = tok[:max_images - seen], target[:max_images - seen]
        if writer is None:
            writer = TokenStoreWriter(out_dir, num_tokens=tok.shape[1], dim=tok.shape[2], shard_images=shard_images, dtype=dtype)
        writer.add(tok.cpu().numpy(), torch.as_tensor(target).cpu().numpy())
        seen += tok.shape[0]
        if log_every and (it + 1) % log_every == 0:
            print(f"  dumped {seen} images", file=sys.stderr)
        if max_images is not None and seen >= max_images:
            break
    if writer is None:
        raise ValueError("dump_tokens: the data loader yielded no batch")
    return writer.close()


def _resolve(spec: str):
    if ":" not in spec:
        raise SystemExit(f"expected module:attribute, got {spec!r}")
    mod, attr = spec.split(":", 1)
    return getattr(importlib.import_module(mod), attr)


def _synthetic_loader(n: int, batch: int, num_tokens: int, dim: int, classes: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    for lo in range(0, n, batch):
        b = min(batch, n - lo)
        yield torch.randn(b, num_tokens, dim, generator=g), torch.randint(0, classes, (b,), generator=g)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m efficient_probing_amd.dump", description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", required=True, help="directory of the token store (created)")
    ap.add_argument("--dtype", default="bf16", choices=sorted(DTYPES), help="stored element type (arithmetic downstream is fp32)")
    ap.add_argument("--shard-images", type=int, default=8192)
    ap.add_argument("--max-images", type=int, default=None)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--encoder", help="module:callable, make_token_fn(device) -> token_fn(images) -> (B, N, D); needs --data")
    src.add_argument("--from-npz", help="a reference dump (tools/dump_tokens.py: keys tokens / images / names)")
    src.add_argument("--synthetic", type=int, metavar="IMAGES", help="N(0,1) tokens and random labels (plumbing, benchmarks)")
    ap.add_argument("--data", help="module:callable, make_loader() -> iterable of (images, ..., targets) batches")
    ap.add_argument("--labels", help="--from-npz: .npy of int labels (default: zeros -- the reference dump carries none)")
    ap.add_argument("--tokens", type=int, default=196, help="--synthetic: tokens per image")
    ap.add_argument("--dim", type=int, default=768, help="--synthetic: channels per token")
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default=None, help="where the encoder runs (default: cuda if available)")
    a = ap.parse_args(argv)
    if a.encoder:
        if not a.data:
            ap.error("--encoder needs --data")
        dev = torch.device(a.device or ("cuda" if torch.cuda.is_available() else "cpu"))
        meta = dump_tokens(_resolve(a.data)(), _resolve(a.encoder)(dev), a.out, dtype=a.dtype, shard_images=a.shard_images,
                           device=dev, max_images=a.max_images, log_every=50)
    elif a.from_npz:
        tok, _ = read_reference_npz(a.from_npz)
        lab = np.load(a.labels) if a.labels else np.zeros(len(tok), np.int64)
        loader = ((torch.from_numpy(tok[lo:lo + a.batch]), torch.from_numpy(lab[lo:lo + a.batch])) for lo in range(0, len(tok), a.batch))
        meta = dump_tokens(loader, lambda x: x, a.out, dtype=a.dtype, shard_images=a.shard_images, max_images=a.max_images)
    else:
        meta = dump_tokens(_synthetic_loader(a.synthetic, a.batch, a.tokens, a.dim, a.classes, a.seed), lambda x: x, a.out,
                           dtype=a.dtype, shard_images=a.shard_images, max_images=a.max_images)
    print(f"wrote {a.out}: {meta['total_images']} images x {meta['num_tokens']} x {meta['dim']} ({meta['dtype']}), "
          f"{len(meta['shards'])} shard(s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
