"""Fused train / eval step of the EP probe head: one C call per phase, no host sync.

``ProbeHeadEngine`` owns the flat fp32 parameter / gradient / optimizer-state buffers of a
``Sequential(EfficientProbing, BatchNorm1d, Linear)`` head (the module's parameters become
views of the flat buffer, so ``state_dict()`` / checkpoints keep working) and runs the body of
the reference hot loop (reference engine_finetune.py:52-77: forward, CE, backward, optimizer)
through ``ep_head_train_step``.

Data parallelism (reference main_linprobe.py:581-583 wraps the model in DDP): every rank holds
a full replica and its own images; per step there is exactly ONE all-reduce (RCCL over xGMI when
the backend is "nccl") of the flat gradient buffer between the backward and the optimizer
phase.  The 1/world averaging is folded into the optimizer's ``inv_scale`` -- no extra pass.
BatchNorm statistics stay per-rank (the reference does not use SyncBN); running stats are
broadcast from rank 0 on ``sync_buffers()`` (eval / checkpoint time).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _native as N
from . import functional as F_

OPTIMIZERS = {"lars": 0, "sgd": 1, "adamw": 2}


class ProbeHeadEngine:
    """Engine of Sequential(EfficientProbing, BatchNorm1d, Linear).  ``CocaHeadEngine`` below reuses everything
    but the four ``_layout`` / ``_new_step`` / ``_ws_bytes`` / ``_call_*`` hooks."""

    def _check_head(self, head):
        from .probe_heads import is_native_ep_head
        if not is_native_ep_head(head):
            raise TypeError("ProbeHeadEngine needs Sequential(EfficientProbing, BatchNorm1d, Linear)")


    def _layout(self):
        """-> (dims struct, parameters in flat order, offsets, total elements)"""
        D = self.pool.v.in_features
        dims = N.EPHeadDims(B=0, N=0, D=D, Q=self.pool.num_queries, d_out=self.pool.d_out, C=self.fc.out_features)
        offs = (C.c_int64 * 4)()
        total = int(self.lib.ep_head_param_offsets(C.byref(dims), offs))
        return dims, [self.pool.cls_token, self.pool.v.weight, self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        return N.EPHeadStep()


    # ---- tables of a resident token store -------------------------------------------------------------------------------
    # Some heads' token passes take per-token LayerNorm statistics / per-image channel statistics.  They are functions of
    # the FROZEN tokens alone, so a resident store computes them once (token_store.ResidentTokenStore.table) and a batch read
    # in place through ``image_index`` looks its rows up instead of re-reading the tokens every step -- which the indexed
    # kernels cannot do at all: without the tables an indexed batch is an argument error.
    _store_kinds = {}          # subclass: struct field -> (store table kind, eps)

    def attach_store(self, store) -> None:
        """Use ``store``'s cached tables whenever a batch is read from it in place (``engine_finetune`` calls this when it
        sees a ``token_store.StoreBatch``).  ``None`` detaches."""
        self._store = store

    def _bind_store_tables(self, xv, image_index) -> None:
        self._bound_tables = {}
        st = getattr(self, "_store", None)
        if st is None or image_index is None or not self._store_kinds:
            return
        # the tables are indexed [image][token] with the STORE's token count: a view with the same base pointer but another
        # token count / row width / batch stride (store.tokens[:, :K]) must not bind them (it would read them with the wrong
        # stride) -- unbound, an indexed batch of the heads that need a table fails loudly instead
        if (xv.data_ptr() != st.tokens.data_ptr() or tuple(xv.shape) != tuple(st.tokens.shape)
                or xv.stride() != st.tokens.stride() or xv.dtype != st.tokens.dtype):
            return
        for field, (kind, eps) in self._store_kinds.items():
            self._bound_tables[field] = st.table(kind, eps)

    def _table_ptr(self, field: str) -> int:
        """Pointer for a step / eval struct: the caller's explicit per-call table, else the attached store's, else 0 (the
        library computes the statistics of the batch itself)."""
        explicit = getattr(self, {"token_stats": "_tokstat", "image_stats": "_imgstat"}.get(field, "_no_such_field"), None)
        if explicit is not None:
            return explicit.data_ptr()
        t = getattr(self, "_bound_tables", {}).get(field)
        return t.data_ptr() if t is not None else 0

    def _ws_bytes(self) -> int:
        return self.lib.ep_head_workspace_bytes(C.byref(self.dims))

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                             self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                             self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                             ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def __init__(self, head: nn.Sequential, optimizer: str = "lars", lr: float = 0.0, weight_decay: float = 0.0,
                 momentum: float = 0.9, trust_coefficient: float = 0.001, betas=(0.9, 0.999), adam_eps: float = 1e-8,
                 process_group=None, loss_scale: float = 1.0, accum_iter: int = 1, broadcast_from_rank0: bool = True,
                 overlap: bool = True, overlap_comm=None, arithmetic: str = "fp32"):
        self._check_head(head)
        if optimizer not in OPTIMIZERS:
            raise ValueError(f"optimizer must be one of {sorted(OPTIMIZERS)}")
        # arithmetic of the step's six contractions (include/ep_hip.h: ep_head_step.arith).  "fp32": fp32 results.
        # "bf16_autocast": what the published runs' ``--amp bfloat16`` does inside autocast (reference engine_finetune.py:52-55)
        # -- operands rounded to bf16, one matrix-core product, fp32 accumulation; softmax, BatchNorm statistics, the loss and
        # the optimizer stay fp32, the token passes keep their arithmetic.  The EP head only; a secondary mode.
        if arithmetic not in ("fp32", "bf16_autocast"):
            raise ValueError("arithmetic must be 'fp32' or 'bf16_autocast'")
        if arithmetic != "fp32" and type(self).__name__ not in ("ProbeHeadEngine", "CocaHeadEngine", "AbmilpHeadEngine"):
            raise ValueError(f"{type(self).__name__}: arithmetic='bf16_autocast' is implemented for the EP, CoCa and AbMILP heads "
                             "(BASELINE configs[3]'s three) only")
        self.arithmetic = arithmetic
        self.head = head
        self.pool, self.bn, self.fc = head[0], head[1], head[2]
        dev = self.fc.weight.device
        if dev.type != "cuda":
            raise RuntimeError("ProbeHeadEngine: the head must be on the GPU (.to('cuda')); there is no CPU path")
        self.device = dev
        self.lib = N.load()
        self.optimizer_name = optimizer
        self.lr, self.weight_decay, self.momentum, self.trust_coefficient = lr, weight_decay, momentum, trust_coefficient
        self.betas, self.adam_eps = betas, adam_eps
        self.loss_scale = float(loss_scale)
        self.accum_iter = int(accum_iter)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.opt_step = 0
        self._micro = 0

        self.dims, self.params_list, self.offsets, self.total = self._layout()
        self.flat_p = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.total, device=dev, dtype=torch.float32)
        n_state = {"lars": 1, "sgd": 0, "adamw": 2}[optimizer]
        self.state = [torch.zeros(self.total, device=dev, dtype=torch.float32) for _ in range(n_state)]
        with torch.no_grad():
            for p, o in zip(self.params_list, self.offsets):
                v = self.flat_p[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self.stats = torch.zeros(4, device=dev, dtype=torch.float32)
        self.found_inf = torch.zeros(1, device=dev, dtype=torch.int32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self._ws = None
        self._ws_key = None
        import os as _os
        if _os.environ.get("EP_AUX_STREAM", "1") == "0":     # diagnostics: everything on one stream
            overlap = False
        self.aux_stream = torch.cuda.Stream(device=dev) if overlap else None
        # Communication overlap (EP head, data parallel): the next step's first token pass needs the updated
        # cls_token only, so the step all-reduces + updates cls_token first and lets the large all-reduce of the
        # other gradients (and their update) run beside the next first token pass.  Same arithmetic, same order.
        # OPT-IN (overlap_comm=True or env EP_OVERLAP_COMM=1) until it has run on two real GPUs: the default data-parallel
        # step is the plain single all-reduce of the flat gradient buffer.  When asked for it needs world > 1, no
        # gradient accumulation and no loss scaling (the inf-skip of a GradScaler needs all gradients before any
        # update); "force": also with one rank (tests).
        import os
        want = overlap_comm if overlap_comm is not None else (os.environ.get("EP_OVERLAP_COMM", "0") == "1")
        self._pipelined = bool(want) and self._supports_comm_overlap() and self.accum_iter == 1 \
            and self.loss_scale == 1.0 and (self.world > 1 or overlap_comm == "force")
        self._pending = None
        # Deferred large update (one rank): the step updates cls_token, then v.weight / fc.* on the aux stream BESIDE the
        # next step's first token pass (which reads cls_token only).  Off by default -- between a train_step() and the
        # next flush() the three large tensors may still be in flight, so only callers that flush() before they read
        # parameters switch it on: engine_finetune.train_one_epoch does for its loop, bench.py does.
        self.defer_update = False
        self._defer_event = None
        self._deferred = False
        self._fs = None                                      # persistent step struct of the one-call step (see _fast_one_call)
        # device-resident GradScaler (attach_scaler): (scaler object, (4,) float tensor {scale, tracker} x 2 slots, slot the next step reads)
        self._scaler = None
        self._scaler_state = None
        self._scaler_slot = 0
        self._fast_ok = type(self).__name__ == "ProbeHeadEngine" and os.environ.get("EP_FAST_STEP", "1") != "0"
        if broadcast_from_rank0 and self.world > 1:
            dist.broadcast(self.flat_p, src=0, group=self.group)     # what DDP does at wrap time
            self.sync_buffers()

    def _supports_comm_overlap(self) -> bool:
        return type(self) is ProbeHeadEngine           # the split phases exist for the EP head only

    # ------------------------------------------------------------------------------------
    def mu_views(self):
        """LARS momentum buffers per parameter (reference state[p]['mu'])."""
        return [self.state[0][o:o + p.numel()].view(p.shape) for p, o in zip(self.params_list, self.offsets)]

    def sync_buffers(self):
        self.flush()
        self.invalidate_planes()          # an epoch boundary: re-split once rather than trust a p.data write nobody could see
        if self.world > 1:
            for b in (self.bn.running_mean, self.bn.running_var, self.bn.num_batches_tracked):
                dist.broadcast(b, src=0, group=self.group)

    def _workspace(self, B: int, Nn: int):
        key = (B, Nn)
        if self._ws_key != key:
            self.dims.B, self.dims.N = B, Nn
            nbytes = self._ws_bytes()
            if nbytes == 0:
                raise RuntimeError(f"head workspace query: {N.last_error()}")
            # zero-filled: the step keeps arrival counters in it that it leaves at zero (include/ep_hip.h, ABI v21)
            self._ws = torch.zeros(nbytes, device=self.device, dtype=torch.uint8)
            self._ws_key = key
            self._planes_token = None
        return self._ws

    # ---- GradScaler state on the device (include/ep_hip.h, ABI v26: ep_head_step.scaler_state) ----
    def attach_scaler(self, scaler) -> None:
        """Drive ``scaler`` (util.misc.NativeScalerWithGradNormCount: torch.cuda.amp.GradScaler semantics, reference
        util/misc.py:260-286) from the fused step: the loss gradient is scaled by its scale, the optimizer unscales, skips the
        update on a non-finite gradient and halves the scale, doubles it after ``growth_interval`` clean steps -- all inside
        the step's kernels, on a 4-float device state; no host read per step.  ``sync_scaler()`` (called by
        ``train_one_epoch`` at the end of an epoch and by ``scaler.state_dict()``) brings the host object up to date."""
        if scaler is None or not getattr(scaler, "_enabled", False):
            self._scaler = self._scaler_state = None
            return
        if self._pipelined or type(self) is not ProbeHeadEngine:
            raise RuntimeError("attach_scaler: the device-resident loss scale is implemented for the EP head's undeferred, "
                               "non-pipelined step (overlap_comm off)")
        if self._scaler is scaler:
            return
        self.flush()
        if self.loss_scale != 1.0:
            raise RuntimeError("attach_scaler: the engine already applies a fixed loss_scale")
        self._scaler = scaler
        sc, tr = float(scaler._scale), float(scaler._growth_tracker)
        self._scaler_state = torch.tensor([sc, tr, sc, tr], device=self.device, dtype=torch.float32)
        self._scaler_slot = 0
        scaler._device_sync = self.sync_scaler                # state_dict() / get_scale() of the host object read through
        scaler._device_push = self.push_scaler

    def push_scaler(self) -> None:
        """The host scaler changed (load_state_dict, a step through the module path): upload its state."""
        if self._scaler is None:
            return
        sc, tr = float(self._scaler._scale), float(self._scaler._growth_tracker)
        self._scaler_state.copy_(torch.tensor([sc, tr, sc, tr], dtype=torch.float32), non_blocking=False)

    def sync_scaler(self) -> None:
        """Copy the device-resident scale / growth tracker back into the attached host scaler (one small read)."""
        if self._scaler is None:
            return
        vals = self._scaler_state.tolist()
        self._scaler._scale = float(vals[2 * self._scaler_slot])
        self._scaler._growth_tracker = int(vals[2 * self._scaler_slot + 1])

    def _scaler_fields(self, s) -> None:
        if self._scaler is None or not hasattr(s, "scaler_state"):
            if hasattr(s, "scaler_state"):
                s.scaler_state = 0
            return
        sc = self._scaler
        s.scaler_state = self._scaler_state.data_ptr()
        s.scaler_slot = self._scaler_slot
        s.scaler_growth = sc._growth_factor; s.scaler_backoff = sc._backoff_factor; s.scaler_interval = sc._growth_interval

    def _step_struct(self, x, bstride, targets, phases, accumulate, lr):
        s = self._new_step()
        s.dims = self.dims
        s.x = x.data_ptr() if x is not None else 0
        s.x_dtype = F_.token_dtype_code(x) if x is not None else N.EP_DTYPE_F32
        s.x_bstride = bstride
        s.targets = targets.data_ptr() if targets is not None else 0
        s.params = self.flat_p.data_ptr(); s.grads = self.flat_g.data_ptr()
        s.opt_state0 = self.state[0].data_ptr() if len(self.state) > 0 else 0
        s.opt_state1 = self.state[1].data_ptr() if len(self.state) > 1 else 0
        s.running_mean = self.bn.running_mean.data_ptr(); s.running_var = self.bn.running_var.data_ptr()
        s.num_batches_tracked = self.bn.num_batches_tracked.data_ptr()
        s.stats = self.stats.data_ptr()
        s.found_inf = self.found_inf.data_ptr(); s.grad_norm = self.grad_norm.data_ptr()
        s.bn_eps = self.bn.eps; s.bn_momentum = self.bn.momentum
        s.grad_scale = self.loss_scale / self.accum_iter
        s.inv_scale = 1.0 / (self.loss_scale * self.world)
        self._scaler_fields(s)
        s.accumulate = int(accumulate)
        s.optimizer = OPTIMIZERS[self.optimizer_name]
        s.lr = self.lr if lr is None else lr
        s.weight_decay = self.weight_decay; s.momentum = self.momentum
        s.trust_coefficient = self.trust_coefficient
        s.beta1, s.beta2 = self.betas; s.adam_eps = self.adam_eps
        s.opt_step = self.opt_step
        s.phases = phases
        s.aux_stream = self.aux_stream.cuda_stream if self.aux_stream is not None else 0
        if hasattr(s, "planes_valid"):
            s.planes_valid = int(self._planes_current())
        if hasattr(s, "arith"):
            s.arith = N.EP_ARITH_BF16_AUTOCAST if getattr(self, "arithmetic", "fp32") == "bf16_autocast" else N.EP_ARITH_F32
        return s

    # ---- bf16 weight planes kept in the workspace (csrc/ep_planes.hip; include/ep_hip.h: ep_head_step.planes_valid) ----
    # A forward phase establishes them for the parameters it sees (it splits them itself when told they are not valid); an
    # optimizer phase over the weight matrices rewrites them as it updates.  They stay valid as long as nobody else writes
    # the parameters: every torch-side write (load_state_dict, broadcast into .data, an optimizer of torch's own ...) bumps
    # the flat buffer's version counter, the library's kernels do not.
    def _param_versions(self):
        """What changes when anybody writes the parameters through torch: the version counters of the Parameters (their
        ``.data`` are views of the flat buffer but each keeps its OWN counter -- ``load_state_dict``, an optimizer of
        torch's own, ``p.copy_`` / ``p.add_`` under no_grad all bump it) and of the flat buffer itself.  Writes through
        ``p.data`` (``p.data.copy_(...)``) are invisible to every version counter: after one, call ``invalidate_planes()``."""
        return (self._ws.data_ptr(), self.flat_p._version) + tuple(p._version for p in self.params_list)

    def invalidate_planes(self) -> None:
        """Make the next step split the weight planes itself (after parameter writes that bypass torch's version counters)."""
        self._planes_token = None

    def _planes_current(self) -> bool:
        tok = getattr(self, "_planes_token", None)
        return (tok is not None and self._ws is not None and type(self) is ProbeHeadEngine and tok == self._param_versions())

    def _planes_after_call(self, phases: int, first_seg: int = 0, num_segs: int = 0) -> None:
        if type(self) is not ProbeHeadEngine or self._ws is None:
            return
        established = self._planes_current()
        if phases & (1 | 8):
            established = True                               # the step made them valid for the parameters it ran on
        # (an optimizer phase never invalidates them: whichever matrices it updates, it rewrites their planes)
        self._planes_token = self._param_versions() if established else None

    # ------------------------------------------------------------------------------------
    def forward_backward(self, x: torch.Tensor, targets: torch.Tensor,
                         image_index: Optional[torch.Tensor] = None) -> None:
        """Phase 1: forward + CE + backward into the flat gradient buffer (accumulating across
        micro-steps when accum_iter > 1).  Adds to ``self.stats``.  With ``image_index`` (int32 (B,)),
        ``x`` is a token store resident in HBM and the batch is read from it in place."""
        self.flush()
        xv, bstride = F_.as_token_view(x)
        _, Nn, D = xv.shape
        iptr, B = F_._index_arg(image_index, xv)
        self._bind_store_tables(xv, image_index)
        ws = self._workspace(B, Nn)
        targets = targets.to(device=self.device, dtype=torch.int64)
        s = self._step_struct(xv, bstride, targets, 1, self._micro > 0, None)
        s.image_index = iptr
        N.check(self._call_train(s, ws), "head train step (fwd+bwd)")
        self._planes_after_call(1)
        self._micro += 1

    def all_reduce_grads(self) -> None:
        """THE collective of a data-parallel step: one sum all-reduce of the flat gradients
        (parallel.all_reduce_flat_grads); the 1/world factor rides on the optimizer's inv_scale."""
        from .parallel import all_reduce_flat_grads
        self.flush()
        all_reduce_flat_grads(self.flat_g, self.group)

    def _optimizer_call(self, lr, first_seg=0, num_segs=0, opt_step=None) -> None:
        ws = self._ws
        if ws is None:
            raise RuntimeError("optimizer_step before any forward_backward")
        s = self._step_struct(None, 0, None, 2, False, lr)
        if opt_step is not None:
            s.opt_step = opt_step
        if num_segs:
            s.opt_first_segment, s.opt_num_segments = first_seg, num_segs
        N.check(self._call_train(s, ws), "head train step (optimizer)")
        if self._scaler is not None:
            self._scaler_slot ^= 1                            # the optimizer phase wrote the other slot
        self._planes_after_call(2, first_seg, num_segs)

    def optimizer_step(self, lr: Optional[float] = None) -> None:
        self.flush()
        self.opt_step += 1
        self._optimizer_call(lr)
        self._micro = 0

    def flush(self) -> None:
        """Complete the deferred half of a pipelined step: wait for the large gradient all-reduce and update the
        tensors it covers.  Called automatically before anything that reads those parameters."""
        if self._deferred:                                    # the current stream waits for the aux-stream update; the host does not
            self._defer_event.wait(torch.cuda.current_stream(self.device))
            self._deferred = False
        if self._pending is not None:
            work, lr, step = self._pending
            self._pending = None
            if work is not None:
                work.wait()                                   # the current stream waits; the host does not
            self._optimizer_call(lr, 1, len(self.params_list) - 1, opt_step=step)

    def _train_step_pipelined(self, x, targets, lr, image_index) -> None:
        xv, bstride = F_.as_token_view(x)
        _, Nn, D = xv.shape
        iptr, B = F_._index_arg(image_index, xv)
        self._bind_store_tables(xv, image_index)
        ws = self._workspace(B, Nn)
        targets = targets.to(device=self.device, dtype=torch.int64)
        s = self._step_struct(xv, bstride, targets, 4, False, None)          # first token pass (needs cls_token only)
        s.image_index = iptr
        N.check(self._call_train(s, ws), "head train step (first token pass)")
        self.flush()                                                         # previous step's large bucket lands here
        s = self._step_struct(xv, bstride, targets, 8, False, None)
        s.image_index = iptr
        N.check(self._call_train(s, ws), "head train step (rest of fwd+bwd)")
        self._planes_after_call(8)
        cut = self.offsets[1]
        w1 = w2 = None
        if self.world > 1 or (dist.is_available() and dist.is_initialized()):
            w1 = dist.all_reduce(self.flat_g[:cut], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w2 = dist.all_reduce(self.flat_g[cut:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w1.wait()
        self.opt_step += 1
        self._optimizer_call(lr, 0, 1)                                       # cls_token: the next step can start
        self._pending = (w2, lr, self.opt_step)

    def _one_call_step(self) -> bool:
        """One rank, no gradient accumulation: nothing happens between backward and update, so the step is ONE library
        call (phases = 3) -- which also lets the library hand the last stage of the cls_token gradient reduction to the
        optimizer's first kernel (one launch less).  Same arithmetic as the two calls."""
        return type(self) is ProbeHeadEngine and self.world == 1 and self.accum_iter == 1 and self._micro == 0

    def _can_defer(self) -> bool:
        import os
        # EP_DEFER_OPT=1 to use it: measured SLOWER on MI355X / ROCm 7.2 (0.452 against 0.433 ms per step at 256x768) -- a
        # dependency between two HIP streams costs 8-12 us of idle queue on each side (EXPERIMENTS.md section 4, round 3)
        return (self.defer_update and self.aux_stream is not None and self.loss_scale == 1.0 and self._scaler is None
                and getattr(self, "arithmetic", "fp32") == "fp32" and os.environ.get("EP_DEFER_OPT", "0") == "1")

    # ---- the one-call step with a PERSISTENT step struct (round 5): the ~40 fields of ep_head_step are written once; a call
    # rewrites the handful that change from step to step (token / target / index pointers, batch geometry, lr, the optimizer's
    # hyper-parameters and step counter, planes_valid) and checks the tokens with plain attribute reads -- 100 -> ~35 us of
    # Python per step (tools/host_overhead.py), which is what a B = 256 .. 512 per-GPU step of the 8-GPU protocol point is
    # measured against (0.2 .. 0.26 ms of device time).  Anything unusual (tokens that need a copy, a dtype conversion, a
    # changed stream) falls back to the general path below.
    def _fast_one_call(self, x, targets, lr, image_index) -> bool:
        if (x.dim() != 3 or not x.is_cuda or targets.dtype is not torch.int64 or targets.device != self.device
                or self._pending is not None or self._deferred):
            return False
        M, Nn, D = x.shape                                   # M = images in the token buffer (a store when image_index is given)
        B = M
        dt = x.dtype
        if dt is torch.float32:
            code, al = N.EP_DTYPE_F32, 4
        elif dt is torch.bfloat16 and D % 8 == 0:
            code, al = N.EP_DTYPE_BF16, 8
        else:
            return False
        st0, st1, st2 = x.stride()
        xp = x.data_ptr()
        if st2 != 1 or st1 != D or (M > 1 and (st0 < Nn * D or st0 % al != 0)) or xp % 16 != 0:
            return False
        if image_index is not None:
            if image_index.dtype is not torch.int32 or not image_index.is_cuda or not image_index.is_contiguous():
                return False
            iptr, B = image_index.data_ptr(), image_index.numel()
        else:
            iptr = 0
        self._bind_store_tables(x, image_index)
        fs = self._fs
        if fs is None:
            fs = self._fs = self._step_struct(None, 0, None, 3, False, None)
            self._fs_ref = C.byref(fs)
        ws = self._ws if self._ws_key == (B, Nn) else self._workspace(B, Nn)
        self.opt_step += 1
        fs.dims.B = B; fs.dims.N = Nn
        # the stride belongs to the BUFFER (as functional.as_token_view): a one-element index batch into a strided store must
        # still step by the store's stride
        fs.x = xp; fs.x_dtype = code; fs.x_bstride = st0 if M > 1 else Nn * D
        fs.image_index = iptr
        fs.targets = targets.data_ptr()
        # the BatchNorm buffers can be rebound between steps (load_state_dict(assign=True), head.to(...)): never step through a
        # cached pointer
        bn = self.bn
        fs.running_mean = bn.running_mean.data_ptr(); fs.running_var = bn.running_var.data_ptr()
        fs.num_batches_tracked = bn.num_batches_tracked.data_ptr()
        fs.bn_eps = bn.eps; fs.bn_momentum = bn.momentum
        fs.lr = self.lr if lr is None else lr
        fs.weight_decay = self.weight_decay; fs.momentum = self.momentum; fs.trust_coefficient = self.trust_coefficient
        fs.beta1, fs.beta2 = self.betas; fs.adam_eps = self.adam_eps
        fs.grad_scale = self.loss_scale; fs.inv_scale = 1.0 / self.loss_scale       # (one rank, no accumulation: _one_call_step)
        fs.opt_step = self.opt_step
        fs.planes_valid = int(self._planes_current())
        if self._scaler is not None or fs.scaler_state:
            self._scaler_fields(fs)
        rc = self.lib.ep_head_train_step(self._fs_ref, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            N.check(rc, "head train step")
        if self._scaler is not None:
            self._scaler_slot ^= 1
        self._planes_token = self._param_versions()          # phases = 3 established / rewrote the planes
        self._micro = 0
        return True

    def _train_step_one_call(self, x, targets, lr, image_index) -> None:
        if self._can_defer():
            return self._train_step_deferred(x, targets, lr, image_index)
        if self._fast_ok and self._fast_one_call(x, targets, lr, image_index):
            return
        self.flush()
        xv, bstride = F_.as_token_view(x)
        _, Nn, D = xv.shape
        iptr, B = F_._index_arg(image_index, xv)
        self._bind_store_tables(xv, image_index)
        ws = self._workspace(B, Nn)
        targets = targets.to(device=self.device, dtype=torch.int64)
        self.opt_step += 1
        s = self._step_struct(xv, bstride, targets, 3, False, lr)
        s.image_index = iptr
        N.check(self._call_train(s, ws), "head train step")
        if self._scaler is not None:
            self._scaler_slot ^= 1
        self._planes_after_call(3)
        self._micro = 0

    def _train_step_deferred(self, x, targets, lr, image_index) -> None:
        """The one-call step with the deferred large update (include/ep_hip.h, phases bits 4 / 5): no flush() in front of
        it -- the library itself makes the stream wait for the previous step's large update behind the first token pass."""
        if self._pending is not None:
            self.flush()
        if self._defer_event is None:
            self._defer_event = torch.cuda.Event()
            self._defer_event.record(torch.cuda.current_stream(self.device))     # (creates the underlying hipEvent_t)
        xv, bstride = F_.as_token_view(x)
        _, Nn, D = xv.shape
        iptr, B = F_._index_arg(image_index, xv)
        self._bind_store_tables(xv, image_index)
        if self._deferred and self._ws_key != (B, Nn):
            self.flush()                                      # the update in flight still uses the old workspace
        ws = self._workspace(B, Nn)
        targets = targets.to(device=self.device, dtype=torch.int64)
        self.opt_step += 1
        s = self._step_struct(xv, bstride, targets, 3 | 16 | (32 if self._deferred else 0), False, lr)
        s.image_index = iptr
        s.defer_event = self._defer_event.cuda_event
        N.check(self._call_train(s, ws), "head train step (deferred update)")
        self._planes_after_call(3)
        self._deferred = True
        self._micro = 0

    def train_step(self, x: torch.Tensor, targets: torch.Tensor, lr: Optional[float] = None,
                   image_index: Optional[torch.Tensor] = None) -> None:
        """One full iteration (accum_iter == 1): forward/backward, gradient all-reduce, update."""
        if self._pipelined:
            return self._train_step_pipelined(x, targets, lr, image_index)
        if self._one_call_step():
            return self._train_step_one_call(x, targets, lr, image_index)
        self.forward_backward(x, targets, image_index)
        if self._micro >= self.accum_iter:
            self.all_reduce_grads()
            self.optimizer_step(lr)

    @torch.no_grad()
    def eval_logits_fp16_autocast(self, x: torch.Tensor, image_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Evaluation logits with the roundings of the reference's evaluation mode.  The reference ALWAYS evaluates
        under ``torch.cuda.amp.autocast()`` (engine_finetune.py:131: fp16 on the GPU): every Linear / matmul takes its
        operands rounded to fp16, accumulates in fp32 and returns fp16; softmax, BatchNorm statistics and the loss stay
        fp32.  Here the same kernels run on operands rounded to fp16 at those points (tokens, scaled queries, both weight
        matrices; pooled projection, BatchNorm output and logits rounded to fp16 on the way out), so a head trained by
        the reference scores like it does there.  Two roundings cannot be placed identically because the kernels pool
        before they project: the reference rounds the (B, Q, N) scores and the per-token values V to fp16; both are
        below the fp16 resolution of the result (tests/test_gpu_parity.py::test_fused_engine_steps_golden pins the distance on the fp16-autocast goldens)."""
        if type(self) is not ProbeHeadEngine:
            return self._eval_logits_fp16_operands(x, image_index)
        self.flush()
        r16 = lambda t: t.to(torch.float16).to(torch.float32)
        # k = x under autocast (ep.py:38,42): the tokens rounded to fp16.  They stay STORED as fp16 -- the forward pass widens them
        # in its ring (EP_DTYPE_F16, exact) -- so fp16 tokens from an autocast backbone are read in place and fp32 ones cost one
        # half-size copy instead of a rounded fp32 copy (round 5).
        if image_index is not None:
            xv, _ = F_.as_token_view(x, allow_f16=True)
            x16 = xv[image_index.long()].to(torch.float16)
        else:
            x16 = x if x.dtype == torch.float16 else x.to(torch.float16)
        if not F_.f16_in_place_ok(x16.shape[-1]):
            x16 = x16.float()                                                # (wide rows: the rounded values through the fp32 kernels)
        q16 = r16(self.pool.cls_token.detach()[0] * self.pool.scale)          # q = cls_token * scale, then the matmul cast
        P, _, _ = F_.pool_forward(x16, q16, 1.0)
        y16 = r16(F_.project_forward(P, r16(self.pool.v.weight.detach())))   # self.v under autocast, attn @ v
        z16 = r16(F_.bn_forward_eval(y16, self.bn.running_mean, self.bn.running_var, self.bn.eps))
        return r16(F_.linear_forward(z16, r16(self.fc.weight.detach()), r16(self.fc.bias.detach())))

    def _norm_parameter_ids(self):
        """ids of the parameters autocast leaves in fp32: those of normalisation layers (layer_norm / batch_norm run in
        fp32 under autocast and take their weights as they are; CoCa's own LayerNorm class, coca_pytorch.py:70-77, too)."""
        ids = set()
        for m in self.head.modules():
            if isinstance(m, (nn.LayerNorm, nn.BatchNorm1d, nn.BatchNorm2d, nn.GroupNorm)) or type(m).__name__ == "LayerNorm":
                ids.update(id(p) for p in m.parameters(recurse=False))
        return ids

    @torch.no_grad()
    def _eval_logits_fp16_operands(self, x: torch.Tensor, image_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reference's evaluation mode (fp16 autocast, engine_finetune.py:131) for the heads whose forward is ONE fused
        library call: the roundings autocast applies to the OPERANDS -- the tokens and every weight a Linear / matmul
        consumes are rounded to fp16 (normalisation weights stay fp32), the fused fp32 forward runs on them and the logits
        are rounded to fp16 on the way out.  The intermediate fp16 roundings inside the pooling (per-token keys / values,
        attention products) are not placed: each is one fp16 ulp of a value that is then averaged over the tokens, below
        the fp16 resolution of the logits.  Pinned on the reference's own fp16-autocast logits (``eval_logits_fp16_autocast``
        of the CoCa and AbMILP fixtures, tests/test_gpu_coca.py / test_gpu_abmilp.py) to a few fp16 ulps of their scale."""
        self.flush()
        r16 = lambda t: t.to(torch.float16).to(torch.float32)
        xv, _ = F_.as_token_view(x)
        if image_index is not None:
            xv = xv[image_index.long()]
        x16 = r16(xv).contiguous()
        keep = self.flat_p.clone()
        skip = self._norm_parameter_ids()
        try:
            for p, o in zip(self.params_list, self.offsets):
                if id(p) not in skip:
                    seg = self.flat_p[o:o + p.numel()]
                    seg.copy_(r16(seg))
            # the base-class forward, NOT the virtual one: the callers that override eval_logits (AbMILP / DINOv2-block / DOLG)
            # have already applied their token selection (`_tokens`: content == "patch" drops token 0, poolings/abmilp.py:56-57)
            # before they come here -- going through their override again would drop a second token
            out = ProbeHeadEngine.eval_logits(self, x16, None, precision="fp32")
        finally:
            self.flat_p.copy_(keep)
        return r16(out)

    @torch.no_grad()
    def eval_logits(self, x: torch.Tensor, image_index: Optional[torch.Tensor] = None,
                    precision: str = "fp32") -> torch.Tensor:
        if precision == "fp16_autocast":
            return self.eval_logits_fp16_autocast(x, image_index)
        if precision != "fp32":
            raise ValueError("precision must be 'fp32' or 'fp16_autocast'")
        self.flush()
        xv, bstride = F_.as_token_view(x, allow_f16=type(self) is ProbeHeadEngine and F_.f16_in_place_ok(x.shape[-1]))   # (ep_head_eval_forward reads fp16 tokens in place)
        _, Nn, D = xv.shape
        iptr, B = F_._index_arg(image_index, xv)
        self._bind_store_tables(xv, image_index)
        ws = self._workspace(B, Nn)
        Cc = self.dims.C
        ldl = F_.padded_ld(Cc)
        out = torch.empty((B, ldl), device=self.device, dtype=torch.float32)
        N.check(self._call_eval(xv, bstride, iptr, out, ldl, ws), "head eval forward")
        return out[:, :Cc]

    def last_train_logits(self) -> torch.Tensor:
        """The train-mode logits (B, C) the last train step computed its loss from (a copy out of the step's workspace;
        diagnostics and tests: the distance of the ``bf16_autocast`` arithmetic from the reference's bf16 head)."""
        fn = {"ProbeHeadEngine": "ep_head_workspace_logits_offset", "CocaHeadEngine": "ep_coca_head_workspace_logits_offset",
              "AbmilpHeadEngine": "ep_abmilp_head_workspace_logits_offset"}.get(type(self).__name__)
        if fn is None or self._ws is None:
            raise RuntimeError("last_train_logits: the EP / CoCa / AbMILP engines after a train step only")
        ldl = C.c_int32(0)
        off = getattr(self.lib, fn)(C.byref(self.dims), C.byref(ldl))      # (self.dims follows the workspace: _workspace)
        if off < 0:
            raise RuntimeError("ep_head_workspace_logits_offset failed")
        B = self._ws_key[0]
        flat = self._ws[off: off + 4 * B * ldl.value].view(torch.float32).view(B, ldl.value)
        return flat[:, : self.dims.C].clone()

    def read_stats(self, reset: bool = True):
        """(mean loss summed over the steps since the last reset, #top-1 hits, #top-5 hits,
        #non-finite rows) -- ONE host sync, at logging frequency rather than per step."""
        vals = self.stats.tolist()
        if reset:
            self.stats.zero_()
        return vals

    def read_stats_async(self):
        """The same four sums WITHOUT draining the queue: they are copied into pinned host memory behind the steps
        enqueued so far (and the device counters cleared behind the copy); ``wait_stats(handle)`` returns them once the
        copy has happened.  A training loop reads window k's handle when it enqueues window k+1's, so logging costs no
        pipeline bubble (a blocking read every 20 iterations costs one step latency: ~5 % at 0.43 ms per step)."""
        if not hasattr(self, "_stats_host"):
            self._stats_host = [torch.empty_like(self.stats, device="cpu").pin_memory() for _ in range(2)]
            self._stats_turn = 0
        host = self._stats_host[self._stats_turn]
        self._stats_turn ^= 1
        host.copy_(self.stats, non_blocking=True)
        self.stats.zero_()
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    @staticmethod
    def wait_stats(handle):
        host, ev = handle
        ev.synchronize()
        return host.tolist()


class CocaHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(CrossAttention (CoCa pooler), BatchNorm1d, Linear): same flat-buffer
    design, data parallelism and call surface as ProbeHeadEngine, through ``ep_coca_head_train_step``."""

    def _check_head(self, head):
        from .probe_heads import is_native_coca_head
        if not is_native_coca_head(head):
            raise TypeError("CocaHeadEngine needs Sequential(poolings.coca.CrossAttention, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        D = p.to_q.in_features
        dims = N.EPCocaDims(B=0, N=0, D=D, H=p.heads, dh=p.dim_head, M=p.img_queries.shape[0],
                            C=self.fc.out_features)
        offs = (C.c_int64 * 7)()
        total = int(self.lib.ep_coca_head_param_offsets(C.byref(dims), offs))
        plist = [p.norm.gamma, p.img_queries, p.to_q.weight, p.to_kv.weight, p.to_out.weight, self.fc.weight,
                 self.fc.bias]
        return dims, plist, list(offs), total

    def _new_step(self):
        s = N.EPCocaStep()
        s.ln_beta = self.pool.norm.beta.data_ptr()
        s.ln_eps = 1e-5                      # F.layer_norm default (coca_pytorch.py:77)
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_coca_head_workspace_bytes(C.byref(self.dims))

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_coca_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(),
                                                N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_coca_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                  self.flat_p.data_ptr(), self.pool.norm.beta.data_ptr(), 1e-5,
                                                  self.bn.running_mean.data_ptr(), self.bn.running_var.data_ptr(),
                                                  self.bn.eps, out.data_ptr(), ldl, ws.data_ptr(), ws.numel(),
                                                  N.current_stream_ptr(self.device))


class AbmilpHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(ABMILPHead, BatchNorm1d, Linear) through ``ep_abmilp_head_train_step``.
    Matrix-core bound (about 11 GFLOP per image per train step at 256 x 1152); tokens must be dense."""

    def _check_head(self, head):
        from .probe_heads import is_native_abmilp_head
        if not is_native_abmilp_head(head):
            raise TypeError("AbmilpHeadEngine needs Sequential(poolings.abmilp.ABMILPHead, BatchNorm1d, Linear)")

    def _layout(self):
        D = self.pool.self_attn.qkv.in_features
        dims = N.EPAbmilpDims(B=0, N=0, D=D, C=self.fc.out_features)
        offs = (C.c_int64 * 9)()
        total = int(self.lib.ep_abmilp_head_param_offsets(C.byref(dims), offs))
        return dims, list(self.pool._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        return N.EPAbmilpStep()

    def _ws_bytes(self) -> int:
        return self.lib.ep_abmilp_head_workspace_bytes(C.byref(self.dims))

    def _tokens(self, x, image_index):
        if image_index is not None:
            # a batch of a resident store: this head's step is a chain of contractions over the token matrix (matrix-core
            # bound, 10 ... 40 x a token read), so the batch is gathered into a contiguous tensor first -- one copy of B x N x D
            x = x.index_select(0, image_index.long())
        if self.pool.content == "patch":
            x = x[:, 1:]
        return F_._contiguous_tokens(x)

    def forward_backward(self, x, targets, image_index=None):
        super().forward_backward(self._tokens(x, image_index), targets, None)

    def eval_logits(self, x, image_index=None, precision="fp32"):
        if precision == "fp16_autocast":
            return self._eval_logits_fp16_operands(self._tokens(x, image_index), None)
        return super().eval_logits(self._tokens(x, image_index), None, precision)

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_abmilp_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(),
                                                  N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_abmilp_head_eval_forward(C.byref(self.dims), xv.data_ptr(), N.EP_DTYPE_F32, bstride,
                                                    self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                    self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                    ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class DinovitHeadEngine(AbmilpHeadEngine):
    """Fused train / eval step of Sequential(DinoViTBlockPooling, BatchNorm1d, Linear) through ``ep_dinovit_head_train_step``.
    Matrix-core bound (one transformer block over every token: about 11.5 GFLOP per image per train step at 256 x 768); tokens
    must be dense fp32."""

    def _check_head(self, head):
        from .probe_heads import is_native_dinovit_head
        if not is_native_dinovit_head(head):
            raise TypeError("DinovitHeadEngine needs Sequential(poolings.dinovit.DinoViTBlockPooling, BatchNorm1d, Linear)")

    def _layout(self):
        b = self.pool.dino_block
        dims = F_.dinovit_dims(0, 0, b.norm1.normalized_shape[0], b.attn.num_heads, b.mlp.fc1.out_features, self.fc.out_features,
                               b.norm1.eps)
        offs = (C.c_int64 * 13)()
        total = int(self.lib.ep_dinovit_head_param_offsets(C.byref(dims), offs))
        return dims, list(self.pool._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        return N.EPDinovitStep()

    def _ws_bytes(self) -> int:
        return self.lib.ep_dinovit_head_workspace_bytes(C.byref(self.dims))

    def _tokens(self, x, image_index):
        if image_index is not None:
            # a batch of a resident store: this head's step is a chain of contractions over the token matrix (matrix-core
            # bound, 10 ... 40 x a token read), so the batch is gathered into a contiguous tensor first -- one copy of B x N x D
            x = x.index_select(0, image_index.long())
        return F_._contiguous_tokens(x)

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_dinovit_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_dinovit_head_eval_forward(C.byref(self.dims), xv.data_ptr(), N.EP_DTYPE_F32, bstride,
                                                     self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                     self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                     ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class SiglipHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(AttentionPoolLatent (SigLIP head), BatchNorm1d, Linear) through
    ``ep_siglip_head_train_step``: the EP token passes with derived queries + proj + residual MLP per image."""

    def _check_head(self, head):
        from .probe_heads import is_native_siglip_head
        if not is_native_siglip_head(head):
            raise TypeError("SiglipHeadEngine needs Sequential(poolings.siglip.AttentionPoolLatent, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = N.EPSiglipDims(B=0, N=0, D=p.q.in_features, H=p.num_heads, hidden=p.mlp.fc1.out_features,
                              C=self.fc.out_features)
        offs = (C.c_int64 * 13)()
        total = int(self.lib.ep_siglip_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        return N.EPSiglipStep()

    def _ws_bytes(self) -> int:
        return self.lib.ep_siglip_head_workspace_bytes(C.byref(self.dims))

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_siglip_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(),
                                                  N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_siglip_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride,
                                                    iptr, self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                    self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                    ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class CaeHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(CAEAttentiveBlock, BatchNorm1d, Linear) through ``ep_cae_head_train_step``
    (LayerNorm-of-tokens mode of the token passes).  ``token_stats`` of a resident store can be passed per call."""

    _store_kinds = {"token_stats": ("token_stats", F_.CAE_LN_EPS)}

    def _check_head(self, head):
        from .probe_heads import is_native_cae_head
        if not is_native_cae_head(head):
            raise TypeError("CaeHeadEngine needs Sequential(poolings.cae.CAEAttentiveBlock, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = N.EPCaeDims(B=0, N=0, D=p.cross_attn.q.in_features, H=p.cross_attn.num_heads, C=self.fc.out_features)
        offs = (C.c_int64 * 16)()
        total = int(self.lib.ep_cae_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPCaeStep()
        s.token_stats = self._table_ptr("token_stats")
        s.ln_eps = F_.CAE_LN_EPS
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_cae_head_workspace_bytes(C.byref(self.dims))

    def train_step(self, x, targets, lr=None, image_index=None, token_stats=None):
        """``token_stats``: (M, N, 2) from functional.token_stats(store), required with ``image_index``."""
        self._tokstat = token_stats
        try:
            super().train_step(x, targets, lr, image_index)
        finally:
            self._tokstat = None

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_cae_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_cae_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                 self._table_ptr("token_stats"),
                                                 F_.CAE_LN_EPS, self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                 self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                 ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class JepaHeadEngine(CaeHeadEngine):
    """Fused train / eval step of Sequential(AttentivePooler (V-JEPA), BatchNorm1d, Linear) through
    ``ep_jepa_head_train_step`` (LayerNorm-of-tokens mode of the token passes)."""

    _store_kinds = {"token_stats": ("token_stats", F_.JEPA_LN_EPS)}

    def _check_head(self, head):
        from .probe_heads import is_native_jepa_head
        if not is_native_jepa_head(head):
            raise TypeError("JepaHeadEngine needs Sequential(poolings.jepa.AttentivePooler, BatchNorm1d, Linear)")

    def _layout(self):
        b = self.pool.cross_attention_block
        dims = N.EPJepaDims(B=0, N=0, D=b.xattn.q.in_features, H=b.xattn.num_heads, hidden=b.mlp.fc1.out_features,
                            C=self.fc.out_features)
        offs = (C.c_int64 * 17)()
        total = int(self.lib.ep_jepa_head_param_offsets(C.byref(dims), offs))
        return dims, list(self.pool._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPJepaStep()
        s.token_stats = self._table_ptr("token_stats")
        s.ln_eps = F_.JEPA_LN_EPS
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_jepa_head_workspace_bytes(C.byref(self.dims))

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_jepa_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_jepa_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                 self._table_ptr("token_stats"),
                                                  F_.JEPA_LN_EPS, self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                  self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                  ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class AimHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(AttentionPoolingClassifier (AIM), BatchNorm1d, Linear) through
    ``ep_aim_head_train_step``.  ``image_stats`` (functional.channel_stats of a resident store, (M, 2, D)) can be passed per
    call: the token BatchNorm's batch statistics are then combined from the cached rows instead of re-reading the tokens."""

    _store_kinds = {"image_stats": ("channel_stats", None)}

    def _check_head(self, head):
        from .probe_heads import is_native_aim_head
        if not is_native_aim_head(head):
            raise TypeError("AimHeadEngine needs Sequential(poolings.aim.AttentionPoolingClassifier, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = N.EPAimDims(B=0, N=0, D=p.k.in_features, H=p.num_heads, C=self.fc.out_features)
        offs = (C.c_int64 * 5)()
        total = int(self.lib.ep_aim_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPAimStep()
        tb = self.pool.bn
        s.image_stats = self._table_ptr("image_stats")
        s.tok_running_mean = tb.running_mean.data_ptr(); s.tok_running_var = tb.running_var.data_ptr()
        s.tok_num_batches_tracked = tb.num_batches_tracked.data_ptr()
        s.tok_bn_eps = tb.eps; s.tok_bn_momentum = tb.momentum
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_aim_head_workspace_bytes(C.byref(self.dims))

    def train_step(self, x, targets, lr=None, image_index=None, image_stats=None):
        self._imgstat = image_stats
        try:
            super().train_step(x, targets, lr, image_index)
        finally:
            self._imgstat = None

    def sync_buffers(self):
        super().sync_buffers()
        if self.world > 1:
            tb = self.pool.bn
            for b in (tb.running_mean, tb.running_var, tb.num_batches_tracked):
                dist.broadcast(b, src=0, group=self.group)

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_aim_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        tb = self.pool.bn
        return self.lib.ep_aim_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                 tb.eps, tb.running_mean.data_ptr(), tb.running_var.data_ptr(),
                                                 self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                 self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                 ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class CaitHeadEngine(CaeHeadEngine):
    """Fused train / eval step of Sequential(CAPooling (CaiT class attention), BatchNorm1d, Linear) through
    ``ep_cait_head_train_step`` (LayerNorm-of-tokens mode of the token passes + the class row as an extra softmax entry)."""

    _store_kinds = {"token_stats": ("token_stats", F_.CAIT_LN_EPS)}

    def _check_head(self, head):
        from .probe_heads import is_native_cait_head
        if not is_native_cait_head(head):
            raise TypeError("CaitHeadEngine needs Sequential(poolings.cait.CAPooling, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = F_.cait_dims(0, 0, p.norm.normalized_shape[0], p.num_heads, p.hidden, self.fc.out_features)
        offs = (C.c_int64 * 23)()
        total = int(self.lib.ep_cait_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPCaitStep()
        s.token_stats = self._table_ptr("token_stats")
        s.ln_eps = F_.CAIT_LN_EPS
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_cait_head_workspace_bytes(C.byref(self.dims))

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_cait_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_cait_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                 self._table_ptr("token_stats"),
                                                  self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                  self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                  ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class ClipHeadEngine(CaeHeadEngine):
    """Fused train / eval step of Sequential(AttentionPool2d (CLIP), BatchNorm1d, Linear) through ``ep_clip_head_train_step``."""

    _store_kinds = {"token_stats": ("token_stats", F_.CLIP_LN_EPS), "xhat_mean": ("xhat_mean", F_.CLIP_LN_EPS)}

    def _check_head(self, head):
        from .probe_heads import is_native_clip_head
        if not is_native_clip_head(head):
            raise TypeError("ClipHeadEngine needs Sequential(poolings.clip.AttentionPool2d, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = F_.clip_dims(0, p.pos_embed.shape[0] - 1, p.norm.normalized_shape[0], p.num_heads, self.fc.out_features)
        offs = (C.c_int64 * 9)()
        total = int(self.lib.ep_clip_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPClipStep()
        s.token_stats = self._table_ptr("token_stats")
        s.xhat_mean = self._table_ptr("xhat_mean") if s.token_stats else 0      # (the table goes with the statistics it was made from)
        s.ln_eps = F_.CLIP_LN_EPS
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_clip_head_workspace_bytes(C.byref(self.dims))

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_clip_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_clip_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                 self._table_ptr("token_stats"),
                                                  self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                  self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl,
                                                  ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))


class DolgHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(SpatialAttention2d (DOLG), BatchNorm1d, Linear) through
    ``ep_dolg_head_train_step``.  Matrix-core bound (4 N D^2 FLOP per image per train step); tokens must be dense fp32."""

    def _check_head(self, head):
        from .probe_heads import is_native_dolg_head
        if not is_native_dolg_head(head):
            raise TypeError("DolgHeadEngine needs Sequential(poolings.dolg.SpatialAttention2d, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = N.EPDolgDims(B=0, N=0, D=p.conv1.in_channels, C=self.fc.out_features)
        offs = (C.c_int64 * 8)()
        total = int(self.lib.ep_dolg_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPDolgStep()
        tb = self.pool.bn
        s.tok_running_mean = tb.running_mean.data_ptr(); s.tok_running_var = tb.running_var.data_ptr()
        s.tok_num_batches_tracked = tb.num_batches_tracked.data_ptr()
        s.tok_bn_eps = tb.eps; s.tok_bn_momentum = tb.momentum
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_dolg_head_workspace_bytes(C.byref(self.dims))

    def _tokens(self, x, image_index):
        if image_index is not None:
            # a batch of a resident store: this head's step is a chain of contractions over the token matrix (matrix-core
            # bound, 10 ... 40 x a token read), so the batch is gathered into a contiguous tensor first -- one copy of B x N x D
            x = x.index_select(0, image_index.long())
        return F_._contiguous_tokens(x)

    def forward_backward(self, x, targets, image_index=None):
        super().forward_backward(self._tokens(x, image_index), targets, None)

    def eval_logits(self, x, image_index=None, precision="fp32"):
        if precision == "fp16_autocast":
            return self._eval_logits_fp16_operands(self._tokens(x, image_index), None)
        return super().eval_logits(self._tokens(x, image_index), None, precision)

    def sync_buffers(self):
        super().sync_buffers()
        if self.world > 1:
            tb = self.pool.bn
            for b in (tb.running_mean, tb.running_var, tb.num_batches_tracked):
                dist.broadcast(b, src=0, group=self.group)

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_dolg_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        tb = self.pool.bn
        return self.lib.ep_dolg_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, tb.eps,
                                                  tb.running_mean.data_ptr(), tb.running_var.data_ptr(), self.flat_p.data_ptr(),
                                                  self.bn.running_mean.data_ptr(), self.bn.running_var.data_ptr(), self.bn.eps,
                                                  out.data_ptr(), ldl, ws.data_ptr(), ws.numel(),
                                                  N.current_stream_ptr(self.device))


class CbamHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(CbamPooling, BatchNorm1d, Linear) through ``ep_cbam_head_train_step`` (streaming
    passes over the tokens).  ``image_stats`` (functional.cbam_channel_table of a resident store, (M, 3, D)) can be passed per
    call: one streaming read per step less."""

    _store_kinds = {"image_stats": ("cbam_channel_table", None)}

    def _check_head(self, head):
        from .probe_heads import is_native_cbam_head
        if not is_native_cbam_head(head):
            raise TypeError("CbamHeadEngine needs Sequential(poolings.cbam.CbamPooling, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = N.EPCbamDims(B=0, N=0, D=p.channel.fc1.in_channels, C=self.fc.out_features, rd=p.rd, ks=p.ks)
        offs = (C.c_int64 * 7)()
        total = int(self.lib.ep_cbam_head_param_offsets(C.byref(dims), offs))
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], list(offs), total

    def _new_step(self):
        s = N.EPCbamStep()
        tb = self.pool.spatial.conv.bn
        s.image_stats = self._table_ptr("image_stats")
        s.tok_running_mean = tb.running_mean.data_ptr(); s.tok_running_var = tb.running_var.data_ptr()
        s.tok_num_batches_tracked = tb.num_batches_tracked.data_ptr()
        s.tok_bn_eps = tb.eps; s.tok_bn_momentum = tb.momentum
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_cbam_head_workspace_bytes(C.byref(self.dims))

    def train_step(self, x, targets, lr=None, image_index=None, image_stats=None):
        self._imgstat = image_stats
        try:
            super().train_step(x, targets, lr, image_index)
        finally:
            self._imgstat = None

    def sync_buffers(self):
        super().sync_buffers()
        if self.world > 1:
            tb = self.pool.spatial.conv.bn
            for b in (tb.running_mean, tb.running_var, tb.num_batches_tracked):
                dist.broadcast(b, src=0, group=self.group)

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_cbam_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        tb = self.pool.spatial.conv.bn
        return self.lib.ep_cbam_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                  self._table_ptr("image_stats"),
                                                  tb.eps, tb.running_mean.data_ptr(), tb.running_var.data_ptr(),
                                                  self.flat_p.data_ptr(), self.bn.running_mean.data_ptr(),
                                                  self.bn.running_var.data_ptr(), self.bn.eps, out.data_ptr(), ldl, ws.data_ptr(),
                                                  ws.numel(), N.current_stream_ptr(self.device))


class SimpoolHeadEngine(ProbeHeadEngine):
    """Fused train / eval step of Sequential(SimPool | SimPool_nolinears, BatchNorm1d, Linear) through
    ``ep_simpool_head_train_step`` (per-image-query token passes).  ``token_stats`` (functional.token_stats(store, 1e-6)) and
    ``image_stats`` (functional.channel_stats(store)) of a resident store can be passed per call; ``token_stats`` is required
    with ``image_index``."""

    _store_kinds = {"token_stats": ("token_stats", F_.SIMPOOL_LN_EPS), "image_stats": ("channel_stats", None)}

    def _check_head(self, head):
        from .probe_heads import is_native_simpool_head
        if not is_native_simpool_head(head):
            raise TypeError("SimpoolHeadEngine needs Sequential(poolings.simpool.SimPool | SimPool_nolinears, BatchNorm1d, Linear)")

    def _layout(self):
        p = self.pool
        dims = N.EPSimpoolDims(B=0, N=0, D=p.norm_patches.normalized_shape[0], H=p.num_heads, C=self.fc.out_features,
                               linears=int(p.linears))
        offs = (C.c_int64 * 6)()
        total = int(self.lib.ep_simpool_head_param_offsets(C.byref(dims), offs))
        offs = list(offs)
        if not p.linears:
            offs = offs[:2] + offs[4:]
        return dims, list(p._tensors()) + [self.fc.weight, self.fc.bias], offs, total

    def _new_step(self):
        s = N.EPSimpoolStep()
        s.token_stats = self._table_ptr("token_stats")
        s.image_stats = self._table_ptr("image_stats")
        s.ln_eps = F_.SIMPOOL_LN_EPS
        return s

    def _ws_bytes(self) -> int:
        return self.lib.ep_simpool_head_workspace_bytes(C.byref(self.dims))

    def train_step(self, x, targets, lr=None, image_index=None, token_stats=None, image_stats=None):
        self._tokstat, self._imgstat = token_stats, image_stats
        try:
            super().train_step(x, targets, lr, image_index)
        finally:
            self._tokstat = self._imgstat = None

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_simpool_head_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_simpool_head_eval_forward(C.byref(self.dims), xv.data_ptr(), F_.token_dtype_code(xv), bstride, iptr,
                                                     self._table_ptr("token_stats"), self._table_ptr("image_stats"), F_.SIMPOOL_LN_EPS, self.flat_p.data_ptr(),
                                                     self.bn.running_mean.data_ptr(), self.bn.running_var.data_ptr(),
                                                     self.bn.eps, out.data_ptr(), ldl, ws.data_ptr(), ws.numel(),
                                                     N.current_stream_ptr(self.device))


class LinearProbeEngine(ProbeHeadEngine):
    """Fused train / eval step of plain linear probing, Sequential(BatchNorm1d, Linear) on one feature vector per
    image (what the registry builds for --cls_features cls / gap / pos ..., reference probe_heads.py:96-99).
    ``x`` is a (B, D) feature matrix; a (B, N, D) token tensor is mean-pooled first by the token pass."""

    def _check_head(self, head):
        from .probe_heads import is_native_lp_head
        if not is_native_lp_head(self._lp_head):
            raise TypeError("LinearProbeEngine needs Sequential(BatchNorm1d(affine=False), Linear)")

    def __init__(self, head, **kw):
        self._lp_head = head
        super().__init__(_LPView(head), **kw)
        self.head = head

    def _layout(self):
        dims = N.EPHeadDims(B=0, N=1, D=self.fc.in_features, Q=1, d_out=1, C=self.fc.out_features)
        offs = (C.c_int64 * 2)()
        total = int(self.lib.ep_lp_param_offsets(C.byref(dims), offs))
        return dims, [self.fc.weight, self.fc.bias], list(offs), total

    def _ws_bytes(self) -> int:
        return self.lib.ep_lp_workspace_bytes(C.byref(self.dims))

    def _features(self, x, image_index):
        if image_index is not None:
            x = x[image_index.long()]
        if x.dim() == 3:
            from .knn import mean_tokens
            x = mean_tokens(x)
        return F_._f32c(x, "features")

    def _workspace(self, B, Nn):
        return super()._workspace(B, 1)

    def forward_backward(self, x, targets, image_index=None):
        f = self._features(x, image_index)
        super().forward_backward(f.view(f.shape[0], 1, f.shape[1]), targets, None)

    def eval_logits(self, x, image_index=None, precision="fp32"):
        f = self._features(x, image_index)
        if precision == "fp16_autocast":
            # reference engine_finetune.py:131 on Sequential(BatchNorm1d, Linear): batch_norm keeps the fp32 features, the
            # Linear takes its input, weight and bias rounded to fp16 and returns fp16
            self.flush()
            r16 = lambda t: t.to(torch.float16).to(torch.float32)
            z = F_.bn_forward_eval(f, self.bn.running_mean, self.bn.running_var, self.bn.eps)
            return r16(F_.linear_forward(r16(z), r16(self.fc.weight.detach()), r16(self.fc.bias.detach())))
        if precision != "fp32":
            raise ValueError("precision must be 'fp32' or 'fp16_autocast'")
        return super().eval_logits(f.view(f.shape[0], 1, f.shape[1]), None)

    def _call_train(self, s, ws) -> int:
        return self.lib.ep_lp_train_step(C.byref(s), ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device))

    def _call_eval(self, xv, bstride, iptr, out, ldl, ws) -> int:
        return self.lib.ep_lp_eval_forward(C.byref(self.dims), xv.data_ptr(), self.flat_p.data_ptr(),
                                           self.bn.running_mean.data_ptr(), self.bn.running_var.data_ptr(), self.bn.eps,
                                           out.data_ptr(), ldl, ws.data_ptr(), ws.numel(),
                                           N.current_stream_ptr(self.device))


class _LPView:
    """Presents Sequential(BN, Linear) with the (pooling, bn, fc) indexing the base engine expects."""

    def __init__(self, head):
        self._h = head

    def __getitem__(self, i):
        return (None, self._h[0], self._h[1])[i]


def make_engine(head: nn.Sequential, **kw) -> ProbeHeadEngine:
    """The fused engine matching a native head: looked up in ``probe_heads.NATIVE_HEADS`` (pooling class -> engine)."""
    from .probe_heads import native_engine_name
    name = native_engine_name(head)
    if name is None:
        return ProbeHeadEngine(head, **kw)           # its constructor says what a native EP head looks like
    return globals()[name](head, **kw)
