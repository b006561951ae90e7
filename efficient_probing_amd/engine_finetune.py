"""Train / eval loops with the reference's call surface (reference engine_finetune.py:22-166):

    train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler,
                    max_norm=0, mixup_fn=None, log_writer=None, args=None) -> {name: global_avg}
    evaluate(data_loader, model, device, *, return_targets_and_preds=False,
             cls_features="cls", return_block=None) -> {"loss", "acc1", "acc5"[, "targets", "preds"]}

Two execution paths, chosen per call:
  * FUSED  -- ``model`` is (or wraps, as ``model.head``) a native ``Sequential(EfficientProbing,
    BatchNorm1d, Linear)`` and the loader yields token tensors ``(B, N, D)`` (live encoder output
    via ``token_fn`` or pre-dumped tokens): one ``ProbeHeadEngine`` step per batch -- forward, loss,
    backward, [one all-reduce], optimizer in the HIP kernels, no host sync inside the step; loss /
    accuracy are read back every ``print_freq`` steps instead of three ``.item()`` per step.
  * MODULE -- anything else: the reference's loop (autocast, criterion, loss scaler, optimizer)
    driving whatever modules ``model`` contains; the native modules run their kernels through
    autograd.  Non-finite loss stops the run with exit code 1 like the reference
    (engine_finetune.py:66-70).
"""
from __future__ import annotations

import math
import sys
import time
from collections import defaultdict, deque
from typing import Callable, Iterable, Optional

import torch

from .util import lr_sched
from .util import misc
from .util.misc import AMP_PRECISIONS
from .token_store import StoreBatch


class SmoothedValue:
    """Windowed / global running average (the subset of reference util/misc.py:22-81 the loops use)."""

    def __init__(self, window_size: int = 20, fmt: str = "{median:.4f} ({global_avg:.4f})"):
        self.deque = deque(maxlen=window_size)
        self.total, self.count, self.fmt = 0.0, 0, fmt

    def update(self, value, n: int = 1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not misc.is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        torch.distributed.barrier()
        torch.distributed.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def median(self):
        return float(torch.tensor(list(self.deque)).median()) if self.deque else 0.0

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def value(self):
        return self.deque[-1] if self.deque else 0.0

    def __str__(self):
        return self.fmt.format(median=self.median, global_avg=self.global_avg, value=self.value)


class MetricLogger:
    def __init__(self, delimiter: str = "  "):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            self.meters[k].update(float(v))

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def __getattr__(self, attr):
        if attr in self.__dict__.get("meters", {}):
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{k}: {v}" for k, v in self.meters.items())

    def log_every(self, iterable, print_freq, header=""):
        t0 = time.time()
        for i, obj in enumerate(iterable):
            yield obj
            if print_freq and i % print_freq == 0 and misc.get_rank() == 0:
                print(f"{header} [{i}] {self}  elapsed {time.time() - t0:.1f}s")


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1, 5)):
    """timm.utils.accuracy: percentage of rows whose target is among the k largest logits."""
    maxk = min(max(topk), output.size(1))
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.size(0) for k in topk]


def _native_head(model):
    from .probe_heads import is_native_head
    m = model.module if hasattr(model, "module") else model
    head = m if is_native_head(m) else getattr(m, "head", None)
    return head if (head is not None and is_native_head(head)) else None


def _optimizer_spec(optimizer):
    """(engine optimizer name, engine keyword arguments) of a torch / drop-in optimizer object (LARS / SGD / AdamW as
    selected by reference main_linprobe.py:403-408)."""
    g = optimizer.param_groups[0]
    cls = type(optimizer).__name__.lower()
    name = "lars" if "lars" in cls else ("adamw" if "adamw" in cls else "sgd")
    kw = dict(lr=g.get("lr", 0.0), weight_decay=g.get("weight_decay", 0.0))
    if name == "lars":
        kw.update(momentum=g.get("momentum", 0.9), trust_coefficient=g.get("trust_coefficient", 0.001))
    if name == "adamw":
        kw.update(betas=tuple(g.get("betas", (0.9, 0.999))), adam_eps=g.get("eps", 1e-8))
    return name, kw


def _alias_optimizer_state(eng, optimizer, name):
    """Keep ``optimizer.state_dict()`` interchangeable with what the reference saves and restores: the optimizer's
    per-parameter state entries become VIEWS of the engine's flat state buffers (LARS: ``mu``, util/lars.py:32-35;
    AdamW: ``exp_avg`` / ``exp_avg_sq`` / ``step`` as torch.optim.AdamW keeps them), and state already present in the
    optimizer (a ``--resume`` ran ``optimizer.load_state_dict`` before the first step) is copied in first."""
    if name == "lars":
        for p, mu in zip(eng.params_list, eng.mu_views()):
            st = optimizer.state[p]
            if "mu" in st:
                mu.copy_(st["mu"])                      # resumed from a checkpoint
            st["mu"] = mu
    elif name == "adamw":
        views = [[b[o:o + p.numel()].view(p.shape) for p, o in zip(eng.params_list, eng.offsets)] for b in eng.state]
        steps = []
        for i, p in enumerate(eng.params_list):
            st = optimizer.state[p]
            if "exp_avg" in st and "exp_avg_sq" in st:
                views[0][i].copy_(st["exp_avg"]); views[1][i].copy_(st["exp_avg_sq"])
                steps.append(int(float(st.get("step", 0))))
            st["exp_avg"], st["exp_avg_sq"] = views[0][i], views[1][i]
            st["step"] = torch.tensor(float(steps[-1]) if steps else 0.0)
        if steps:
            if len(set(steps)) != 1:
                raise RuntimeError(f"AdamW resume: the head's parameters carry different step counts {sorted(set(steps))}")
            eng.opt_step = steps[0]                     # the bias correction continues where the checkpoint stopped

        def _publish_step(opt):                         # state_dict() / checkpoint: the engine owns the step count
            for p in eng.params_list:
                opt.state[p]["step"] = torch.tensor(float(eng.opt_step))
        optimizer.register_state_dict_pre_hook(_publish_step)


def get_engine(model, optimizer=None, args=None):
    """The fused engine attached to ``model``.  The TRAINING engine is created the first time an optimizer is passed,
    from that optimizer's class and hyper-parameters; a call without an optimizer before that (``evaluate()`` first)
    gets a stateless evaluation engine that is replaced, not reused, when training starts.  A later call with another
    optimizer class raises; hyper-parameters are re-read from the optimizer's first group on every call."""
    from .engine import make_engine
    m = model.module if hasattr(model, "module") else model
    eng = getattr(m, "_ep_engine", None)
    if eng is not None and optimizer is not None:
        name, kw = _optimizer_spec(optimizer)
        if getattr(eng, "_eval_only", False):
            eng = None                                  # built by evaluate() before training: never train on it
        elif name != eng.optimizer_name:
            raise RuntimeError(f"the fused engine of this model was built for {eng.optimizer_name}; got a "
                               f"{type(optimizer).__name__} optimizer (build a new model / head for another optimizer)")
        else:
            eng.invalidate_planes()                     # re-entry (every epoch): parameter writes through p.data are invisible
            eng.weight_decay = kw["weight_decay"]       # to the version counters the cached weight planes are checked against
            if name == "lars":
                eng.momentum, eng.trust_coefficient = kw["momentum"], kw["trust_coefficient"]
            if name == "adamw":
                eng.betas, eng.adam_eps = kw["betas"], kw["adam_eps"]
    if eng is None:
        head = _native_head(model)
        if head is None:
            return None
        accum = getattr(args, "accum_iter", 1) if args is not None else 1
        if optimizer is None:
            eng = make_engine(head, optimizer="sgd", accum_iter=accum)     # no optimizer state: evaluation only
            eng._eval_only = True
        else:
            name, kw = _optimizer_spec(optimizer)
            # arithmetic of the EP step's contractions: fp32 unless asked for.  The reference's published command lines pass
            # ``--amp bfloat16`` (README.md:639-645; engine_finetune.py:52-55 runs the head under autocast); here that flag keeps
            # the fp32 arithmetic (a superset in accuracy) and the bf16 single-product mode is an explicit opt-in:
            # ``args.ep_arithmetic = "bf16_autocast"`` or EP_ARITHMETIC=bf16_autocast (the EP head only).
            import os
            arith = getattr(args, "ep_arithmetic", None) or os.environ.get("EP_ARITHMETIC", "fp32")
            if arith != "fp32":
                kw = dict(kw, arithmetic=arith)
            eng = make_engine(head, optimizer=name, accum_iter=accum, **kw)
            eng._eval_only = False
            _alias_optimizer_state(eng, optimizer, name)
        m._ep_engine = eng
    return eng


def _plain_cross_entropy(criterion) -> bool:
    """What the fused step computes: mean cross-entropy over the batch, no class weights, no label smoothing
    (reference main_linprobe.py:589)."""
    return criterion is None or (type(criterion) is torch.nn.CrossEntropyLoss and criterion.weight is None
                                 and criterion.label_smoothing == 0.0 and criterion.reduction == "mean"
                                 and criterion.ignore_index == -100)


def _attach_store(engine, batch) -> None:
    """A batch of a resident token store: let the engine look up the store's cached per-token / per-image tables (LayerNorm
    statistics of the tokens, channel statistics -- ``ResidentTokenStore.table``) through the batch's ``image_index``.  The
    heads whose token passes take such tables cannot read an indexed batch without them."""
    owner = getattr(batch, "owner", None) if isinstance(batch, StoreBatch) else None
    if engine is not None and owner is not None and getattr(engine, "_store", None) is not owner and hasattr(engine, "attach_store"):
        engine.attach_store(owner)


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable,
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int, loss_scaler,
                    max_norm: float = 0, mixup_fn=None, log_writer=None, args=None,
                    token_fn: Optional[Callable] = None):
    """``token_fn(samples) -> (B, N, D) tokens`` (the frozen encoder) enables the fused path for image
    loaders; loaders that already yield 3-D token tensors take it directly.  A loader may also yield
    ``token_store.StoreBatch(store_tensor, image_index, targets)`` (``token_store.ResidentTokenStore.loader``): the batch
    is then read in place from the HBM-resident store through ``image_index`` -- no gather copy.  Any other batch is read
    as the reference reads it: ``batch[0]`` the samples, ``batch[-1]`` the targets (engine_finetune.py:40-41 there)."""
    model.train(True)
    metric_logger = MetricLogger()
    metric_logger.add_meter("lr", SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = f"Epoch: [{epoch}]"
    print_freq = 20
    accum_iter = getattr(args, "accum_iter", 1)
    amp = getattr(args, "amp", "none")
    n_iter = len(data_loader)
    # the fused step is plain CE + the optimizer: anything it would silently drop (mixup, gradient clipping, another
    # criterion) takes the module path, which honours it
    fusable = mixup_fn is None and not max_norm and _plain_cross_entropy(criterion)
    engine = get_engine(model, optimizer, args) if fusable else None
    if engine is not None and hasattr(engine, "defer_update"):
        # inside this loop nothing reads the head's parameters between two steps, so the engine MAY let the large update of a
        # step run beside the next step's first token pass (flushed below, before anything can read them) -- it only does
        # with EP_DEFER_OPT=1: measured slower on this stack (engine._can_defer, EXPERIMENTS.md section 4 round 3)
        engine.defer_update = True
    # the reference steps its GradScaler in every iteration, bf16 and fp32 runs included (util/misc.py:260-286): the fused step
    # does the same on a device-resident state (scale(loss), unscale, skip on inf / nan, update) -- the host object is brought up to
    # date at the end of the epoch and whenever its state_dict() / get_scale() is asked for
    if engine is not None and hasattr(engine, "attach_scaler") and hasattr(loss_scaler, "_growth_tracker"):
        try:
            engine.attach_scaler(loss_scaler)
        except RuntimeError:
            pass                                  # (pipelined / other heads: the fused step keeps its fixed scale, as before)
    optimizer.zero_grad()
    pending = pending_images = 0                  # fused steps (and their images) whose statistics are still on the GPU

    reads = []                                    # (handle, steps) of windows whose statistics are on their way to the host

    def account(vals, n_steps, n_images):
        loss_sum, top1, top5, bad = vals
        if bad > 0 or not math.isfinite(loss_sum):
            print(f"Loss is non-finite ({loss_sum}), stopping training")
            sys.exit(1)
        metric_logger.meters["loss"].update(loss_sum / n_steps, n_steps)
        metric_logger.meters["acc1"].update(top1 * 100.0 / n_images, n_steps)
        metric_logger.meters["acc5"].update(top5 * 100.0 / n_images, n_steps)

    def flush_stats(n_steps, n_images, last=False):
        # a window's sums are read when the NEXT window's read-back is enqueued (the last one at the end of the epoch), so
        # the meters -- and the non-finite check, reference engine_finetune.py:62-64 -- run one window behind the queue
        # instead of draining it; engines without the asynchronous form read in place
        # (the image count of a window is recorded WITH it: a last partial batch must not change an earlier window's mean)
        if not hasattr(engine, "read_stats_async"):
            return account(engine.read_stats(), n_steps, n_images)
        reads.append((engine.read_stats_async(), n_steps, n_images))
        while len(reads) > (0 if last else 1):
            handle, n, ni = reads.pop(0)
            account(engine.wait_stats(handle), n, ni)

    for step, batch in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        samples, targets = batch[0], batch[-1]
        image_index = batch[1] if isinstance(batch, StoreBatch) else None      # resident token store, read in place
        _attach_store(engine, batch)
        if step % accum_iter == 0:
            lr = lr_sched.adjust_learning_rate(optimizer, step / n_iter + epoch, args)
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        fused = engine is not None and (samples.dim() == 3 or token_fn is not None)
        if image_index is not None and not fused:
            raise RuntimeError("(store, image_index, targets) batches need the fused engine path (a native head, plain "
                               "cross-entropy, no mixup / gradient clipping)")
        if fused:
            tokens = samples if samples.dim() == 3 else token_fn(samples)
            if accum_iter == 1:        # one call: lets a data-parallel EP step overlap its large all-reduce
                engine.train_step(tokens.detach(), targets, lr=max(g["lr"] for g in optimizer.param_groups),
                                  image_index=image_index)
            else:
                engine.forward_backward(tokens.detach(), targets, image_index)
                if (step + 1) % accum_iter == 0:
                    engine.all_reduce_grads()
                    engine.optimizer_step(lr=max(g["lr"] for g in optimizer.param_groups))
            pending += 1
            pending_images += int(targets.shape[0])
            if pending == print_freq or step == n_iter - 1:
                flush_stats(pending, pending_images, last=step == n_iter - 1)
                pending = pending_images = 0
        else:
            if mixup_fn is not None:
                samples, targets = mixup_fn(samples, targets)
            with torch.autocast("cuda", enabled=(amp != "none" and samples.is_cuda), dtype=AMP_PRECISIONS[amp]):
                outputs = model(samples)
                loss = criterion(outputs, targets)
            acc1, acc5 = accuracy(outputs.float(), targets if targets.dim() == 1 else targets.argmax(1))
            metric_logger.update(acc1=acc1.item(), acc5=acc5.item())
            loss_value = loss.item()
            if not math.isfinite(loss_value):
                print(f"Loss is {loss_value}, stopping training")
                sys.exit(1)
            loss = loss / accum_iter
            loss_scaler(loss, optimizer, clip_grad=max_norm if max_norm else None, parameters=model.parameters(),
                        create_graph=False, update_grad=(step + 1) % accum_iter == 0)
            if (step + 1) % accum_iter == 0:
                optimizer.zero_grad()
            metric_logger.update(loss=loss_value)
        metric_logger.update(lr=max(g["lr"] for g in optimizer.param_groups))
        if log_writer is not None and (step + 1) % accum_iter == 0 and "loss" in metric_logger.meters:
            epoch_1000x = int((step / n_iter + epoch) * 1000)
            log_writer.add_scalar("loss", misc.all_reduce_mean(metric_logger.meters["loss"].value), epoch_1000x)
            log_writer.add_scalar("lr", metric_logger.meters["lr"].value, epoch_1000x)
    if pending:                           # (a loader shorter than its len(): steps not yet enqueued for read-back)
        flush_stats(pending, pending_images, last=True)
    while reads:                          # ... and the window still on its way
        handle, n, ni = reads.pop(0)
        account(engine.wait_stats(handle), n, ni)
    eng = getattr(model.module if hasattr(model, "module") else model, "_ep_engine", None)
    if eng is not None:
        eng.flush()                       # a pipelined / deferred step leaves its large update half a step behind
        if hasattr(eng, "sync_scaler"):
            eng.sync_scaler()
        if hasattr(eng, "defer_update"):
            eng.defer_update = False
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: m.global_avg for k, m in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(data_loader, model, device, *, return_targets_and_preds: bool = False, cls_features: str = "cls",
             return_block: Optional[int] = None, token_fn: Optional[Callable] = None, precision: Optional[str] = None):
    """``precision``: "fp16_autocast" -- the DEFAULT, because it is what the reference does: it always evaluates under
    ``torch.cuda.amp.autocast()`` (engine_finetune.py:131), so a drop-in must not change the numbers silently -- or "fp32"
    (the kernels' own arithmetic, on request).  The environment variable EP_EVAL_PRECISION overrides the default for callers
    that cannot pass the argument (the reference's main_linprobe.py call sites).  The fp16 mode is reproduced on the fused
    path by ``engine.eval_logits(..., precision="fp16_autocast")`` (EP and plain linear probing: every autocast rounding
    placed; the other heads: operands and logits rounded, see ``ProbeHeadEngine._eval_logits_fp16_operands``) and on the
    module path by torch's autocast itself."""
    if precision is None:
        import os
        precision = os.environ.get("EP_EVAL_PRECISION", "fp16_autocast")
    if precision not in ("fp32", "fp16_autocast"):
        raise ValueError("precision must be 'fp32' or 'fp16_autocast'")
    model.eval()
    metric_logger = MetricLogger()
    engine = get_engine(model)
    if engine is not None:
        engine.sync_buffers()                      # rank 0's running statistics, as DDP's buffer broadcast gives
    all_t, all_p = [], []
    for batch in metric_logger.log_every(data_loader, 10, "Test:"):
        images, target = batch[0].to(device, non_blocking=True), batch[-1].to(device, non_blocking=True)
        index = batch[1] if isinstance(batch, StoreBatch) else None  # resident token store, read in place
        _attach_store(engine, batch)
        if index is not None and engine is None:
            raise RuntimeError("(store, image_index, targets) batches need a native head (the fused engine path)")
        if engine is not None and (images.dim() == 3 or token_fn is not None):
            output = engine.eval_logits(images if images.dim() == 3 else token_fn(images), image_index=index, precision=precision)
        else:
            with torch.autocast("cuda", enabled=(precision == "fp16_autocast" and images.is_cuda), dtype=torch.float16):
                output = model(images)
            output = output.float()
        loss = torch.nn.functional.cross_entropy(output, target)
        acc1, acc5 = accuracy(output, target)
        n = target.shape[0]
        metric_logger.update(loss=loss.item())
        metric_logger.meters["acc1"].update(acc1.item(), n=n)
        metric_logger.meters["acc5"].update(acc5.item(), n=n)
        if return_targets_and_preds:
            all_t.append(target.cpu())
            all_p.append(output.argmax(1).cpu())
    metric_logger.synchronize_between_processes()
    stats = {k: m.global_avg for k, m in metric_logger.meters.items()}
    print("* Acc@1 {:.3f} Acc@5 {:.3f} loss {:.3f}".format(stats.get("acc1", 0), stats.get("acc5", 0), stats.get("loss", 0)))
    if return_targets_and_preds:
        stats["targets"], stats["preds"] = torch.cat(all_t), torch.cat(all_p)
    return stats


@torch.no_grad()
def extract_features(data_loader, model, device, *, return_targets_and_preds: bool = False, cls_features: str = "cls",
                     return_block: Optional[int] = None, token_fn: Optional[Callable] = None):
    """One feature vector per image for the k-NN evaluation -- signature and result of the reference's
    ``extract_features`` (engine_finetune.py:168-222 there): ``stats`` (the meters: none are updated, as there) plus
    ``stats["targets"]`` / ``stats["features"]`` (CPU tensors) when ``return_targets_and_preds``.

    Where the features come from, in this order:
      * a ``token_store.StoreBatch`` -- the batch's tokens are read in place from the resident store and averaged over the
        token axis by the streaming token pass (``knn.mean_tokens``; the reference takes ``output_feat.mean(dim=1)`` of 3-D
        wrapper outputs, :205-206);
      * ``token_fn(images)`` (the frozen encoder) when given;
      * otherwise the model itself, called the way the reference calls its ViT wrappers:
        ``model.forward(images, return_features=cls_features, return_block=return_block, return_backbone_features=True)``
        -> ``(_, features)`` (:197).
    3-D outputs are mean-pooled (on the GPU by the token pass), higher ranks flattened (:207-208)."""
    from .knn import mean_tokens
    metric_logger = MetricLogger()
    if model is not None:
        model.eval()
    targets, features = [], []
    for batch in metric_logger.log_every(data_loader, 10, "Test:"):
        images, target = batch[0], batch[-1]
        target = target.to(device, non_blocking=True)
        if isinstance(batch, StoreBatch):
            feat = mean_tokens(images, image_index=batch[1])
        else:
            images = images.to(device, non_blocking=True)
            if token_fn is not None:
                feat = token_fn(images)
            elif images.dim() == 3:                       # a loader of pre-extracted token tensors
                feat = images
            else:
                m = model.module if hasattr(model, "module") else model
                with torch.autocast("cuda", enabled=images.is_cuda, dtype=torch.float16):   # reference :189
                    _, feat = m.forward(images, return_features=cls_features, return_block=return_block,
                                        return_backbone_features=True)
            if feat.dim() == 3:
                feat = mean_tokens(feat if feat.dtype in (torch.float32, torch.bfloat16) else feat.float())
            elif feat.dim() > 3:
                feat = feat.flatten(1)
        targets.append(target.cpu())
        features.append(feat.float().cpu())
    metric_logger.synchronize_between_processes()
    stats = {k: m.global_avg for k, m in metric_logger.meters.items()}
    if return_targets_and_preds:
        stats["targets"] = torch.cat(targets)
        stats["features"] = torch.cat(features)
    return stats
