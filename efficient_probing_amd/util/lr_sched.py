"""Per-iteration learning-rate schedule of the probing protocol: linear warm-up for
``args.warmup_epochs`` epochs, then a half cosine down to ``args.min_lr`` at ``args.epochs``
(behaviour of reference util/lr_sched.py:3-15; fractional ``epoch`` = step / len(loader) + epoch,
engine_finetune.py:43-44).  Host-side float64, exactly like the reference."""
import math


def lr_at(epoch: float, base_lr: float, min_lr: float, warmup_epochs: float, total_epochs: float) -> float:
    if epoch < warmup_epochs:
        return base_lr * epoch / warmup_epochs
    progress = (epoch - warmup_epochs) / (total_epochs - warmup_epochs)
    return min_lr + (base_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * progress))


def adjust_learning_rate(optimizer, epoch, args):
    """Write the scheduled lr into every param group (scaled by the group's ``lr_scale`` when it
    has one) and return the unscaled value."""
    lr = lr_at(epoch, args.lr, args.min_lr, args.warmup_epochs, args.epochs)
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr


def absolute_lr(blr: float, eff_batch_size: int) -> float:
    """``args.lr = args.blr * eff_batch_size / 256`` (reference main_linprobe.py:572-573)."""
    return blr * eff_batch_size / 256
