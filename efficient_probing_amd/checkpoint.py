"""Checkpoint / export compatibility with the reference.

Same on-disk dictionaries as reference util/misc.py:304-393 (``save_model`` / ``load_model``) and
tools/export_ep_heads.py:124-138, so heads trained by either code base load in the other:

  checkpoint-*.pth : {'saved_module': 'head', 'model': head.state_dict(), 'optimizer': optimizer.state_dict(),
                      'epoch', 'scaler', 'args', 'test_stats'}
      head keys: 0.cls_token 0.v.weight 1.running_mean 1.running_var 1.num_batches_tracked 2.weight 2.bias
      LARS state: per-parameter {'mu': tensor}
  ep_head.pth      : {'state_dict': <head keys>, 'meta': {...}}  (+ config.json, manifest.json)
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Optional

import torch


def _is_main() -> bool:
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, test_stats,
               include_epoch_in_filename: bool = True, filename_tag: Optional[str] = None):
    """``model_without_ddp`` is what gets saved: the probe head for a frozen-backbone run (reference
    main_linprobe.py:641-659 passes ``model_without_ddp.head``)."""
    suffix = getattr(args, "suffix", "")
    name = filename_tag if filename_tag is not None else (f"{suffix}_{epoch}" if include_epoch_in_filename else suffix)
    path = Path(args.output_dir) / f"checkpoint-{name}.pth"
    payload = {
        "saved_module": getattr(model_without_ddp, "_ep_saved_module", "head"),
        "model": {k: v.detach().cpu().clone() for k, v in model_without_ddp.state_dict().items()},
        "optimizer": optimizer.state_dict(),
        "epoch": epoch,
        "scaler": loss_scaler.state_dict() if loss_scaler is not None else {},
        "args": args,
        "test_stats": test_stats,
    }
    if _is_main():
        os.makedirs(args.output_dir, exist_ok=True)
        torch.save(payload, path)
    return path


def load_model(args, model_without_ddp, optimizer=None, loss_scaler=None, strict: bool = True):
    """Resume semantics of the reference: head-only checkpoints are routed into ``model.head``; a strict
    failure falls back to non-strict; a resume that matches zero tensors raises instead of silently
    continuing with an untrained head (reference util/misc.py:346-379)."""
    if not getattr(args, "resume", ""):
        return None
    ck = torch.load(args.resume, map_location="cpu", weights_only=False)
    sd = ck["model"] if "model" in ck else ck["state_dict"]          # exported heads use 'state_dict'
    target = model_without_ddp
    own = set(model_without_ddp.state_dict())
    if (ck.get("saved_module") == "head" or not (set(sd) & own)) and hasattr(model_without_ddp, "head"):
        if set(sd) & set(model_without_ddp.head.state_dict()):
            target = model_without_ddp.head
    try:
        target.load_state_dict(sd, strict=strict)
    except RuntimeError:
        if not strict:
            raise
        missing, unexpected = target.load_state_dict(sd, strict=False)
        loaded = len(set(sd) & set(target.state_dict()))      # tensors that actually found a home
        if loaded == 0:
            raise RuntimeError(f"resume from {args.resume} matched 0 of {len(sd)} checkpoint tensors against "
                               f"{type(target).__name__}; refusing to continue with an untrained head")
    resuming = optimizer is not None and "optimizer" in ck and "epoch" in ck and not getattr(args, "eval", False) \
        and not getattr(args, "knn_eval", False)
    if resuming:
        optimizer.load_state_dict(ck["optimizer"])
        args.start_epoch = ck["epoch"] + 1
        if loss_scaler is not None and ck.get("scaler"):
            loss_scaler.load_state_dict(ck["scaler"])
        return ck.get("test_stats")
    return None


def export_head(head: torch.nn.Module, meta: dict, out_dir: str, slug: str) -> str:
    """Release format of reference tools/export_ep_heads.py:124-138."""
    dst = os.path.join(out_dir, slug)
    os.makedirs(dst, exist_ok=True)
    sd = {k: v.detach().cpu().clone() for k, v in head.state_dict().items()}
    torch.save({"state_dict": sd, "meta": meta}, os.path.join(dst, "ep_head.pth"))
    with open(os.path.join(dst, "config.json"), "w") as f:
        json.dump(meta, f, indent=1)
    entry = {**meta, "params_incl_bn_stats": int(sum(v.numel() for v in sd.values())), "file": slug + "/ep_head.pth",
             "size_mb": round(os.path.getsize(os.path.join(dst, "ep_head.pth")) / 1e6, 1)}
    mpath = os.path.join(out_dir, "manifest.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {"heads": [], "missing": []}
    manifest["heads"] = [h for h in manifest["heads"] if h.get("file") != entry["file"]] + [entry]
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1)
    return dst


def load_exported_head(head: torch.nn.Module, path: str, strict: bool = True) -> dict:
    ck = torch.load(path, map_location="cpu", weights_only=False)
    head.load_state_dict(ck["state_dict"], strict=strict)
    return ck.get("meta", {})
