"""LARS for the probe head, stepping in one fused HIP kernel pair over a flat parameter buffer.

Interface and numerics of reference util/lars.py:4-37: ``LARS(params, lr=0, weight_decay=0,
momentum=0.9, trust_coefficient=0.001)``; tensors with ``ndim > 1`` get weight decay and the
trust ratio ``tc * ||p|| / ||g + wd p||`` (1 when either norm is 0), 1-D tensors get neither;
``mu = momentum * mu + dp ; p -= lr * mu``; the momentum buffer is ``state[p]['mu']`` so
optimizer checkpoints are interchangeable with the reference's.

MI355X-native layout: on the first step all parameters of a group are moved into ONE flat fp32
buffer (``p.data`` become views of it), with flat gradient and momentum buffers of the same
layout -- the optimizer is then two kernel launches per group regardless of the number of
tensors, and the flat gradient buffer is what a data-parallel run all-reduces once per step.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import functional as F_


class _FlatGroup:
    """Flat param / grad / state buffers for one param group."""

    def __init__(self, params: List[torch.nn.Parameter], n_state: int):
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("native optimizers need the parameters on the GPU (no CPU path); "
                               f"got device {dev}")
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("native optimizers need fp32 parameters on one device")
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4           # every tensor starts 16-byte aligned
        self.total = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.state = [torch.zeros(off, device=dev, dtype=torch.float32) for _ in range(n_state)]
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                self.view(self.flat_p, p, o).copy_(p.data)
                p.data = self.view(self.flat_p, p, o)
        self.segments = F_.make_segments([(o, p.numel(), p.ndim > 1) for p, o in zip(params, self.offsets)])
        self.workspace = F_.optim_workspace(self.total, len(params), dev)
        self.found_inf = torch.zeros(1, device=dev, dtype=torch.int32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)

    @staticmethod
    def view(flat, p, off):
        return flat[off:off + p.numel()].view(p.shape)

    def grad_view(self, i):
        return self.view(self.flat_g, self.params[i], self.offsets[i])

    def state_view(self, k, i):
        return self.view(self.state[k], self.params[i], self.offsets[i])

    def gather_grads(self) -> bool:
        """Make flat_g hold the current gradients (copying any that autograd allocated
        elsewhere) and re-point ``p.grad`` at the flat views.  Returns False if no parameter
        has a gradient."""
        any_grad = False
        for i, p in enumerate(self.params):
            gv = self.grad_view(i)
            if p.grad is None:
                gv.zero_()
                continue
            any_grad = True
            if p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv
        return any_grad


class LARS(torch.optim.Optimizer):
    """LARS optimizer, no rate scaling or weight decay for parameters <= 1D."""

    def __init__(self, params, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001):
        defaults = dict(lr=lr, weight_decay=weight_decay, momentum=momentum,
                        trust_coefficient=trust_coefficient)
        super().__init__(params, defaults)
        self._flat: List[Optional[_FlatGroup]] = [None] * len(self.param_groups)
        self.last_found_inf = None
        self.last_grad_norm = None

    def _group(self, gi: int) -> _FlatGroup:
        if self._flat[gi] is None:
            ps = [p for p in self.param_groups[gi]["params"] if p.requires_grad]
            fg = _FlatGroup(ps, n_state=1)
            for i, p in enumerate(ps):
                st = self.state[p]
                mu = fg.state_view(0, i)
                if "mu" in st:                              # resumed from a checkpoint
                    mu.copy_(st["mu"])
                st["mu"] = mu
            self._flat[gi] = fg
        return self._flat[gi]

    def flat_grad_buffers(self) -> List[torch.Tensor]:
        """The flat gradient buffer of every group (what a DP run all-reduces)."""
        return [self._group(gi).flat_g for gi in range(len(self.param_groups))]

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the flat gradient buffers in place and keeps ``p.grad`` pointing into them, so
        autograd accumulates straight into the buffer the kernel reads."""
        for gi in range(len(self.param_groups)):
            fg = self._flat[gi]
            if fg is None:
                for p in self.param_groups[gi]["params"]:
                    p.grad = None
                continue
            fg.flat_g.zero_()
            for i, p in enumerate(fg.params):
                p.grad = fg.grad_view(i)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for gi, fg in enumerate(self._flat):
            if fg is None:
                continue
            for i, p in enumerate(fg.params):
                st = self.state[p]
                mu = fg.state_view(0, i)
                if "mu" in st and st["mu"].data_ptr() != mu.data_ptr():
                    mu.copy_(st["mu"])
                st["mu"] = mu

    @torch.no_grad()
    def step(self, closure=None, inv_scale: float = 1.0):
        """One update.  ``inv_scale`` (1 / loss scale) is applied to the gradients inside the
        kernel; if any unscaled gradient is non-finite nothing is updated and
        ``last_found_inf`` (device int32) is 1 -- the GradScaler contract."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, g in enumerate(self.param_groups):
            fg = self._group(gi)
            if not fg.gather_grads():
                continue
            F_.lars_step(fg.flat_p, fg.flat_g, fg.state[0], fg.segments, float(g["lr"]),
                         float(g["weight_decay"]), float(g["momentum"]), float(g["trust_coefficient"]),
                         float(inv_scale), fg.found_inf, fg.grad_norm, fg.workspace)
            self.last_found_inf = fg.found_inf
            self.last_grad_norm = fg.grad_norm
        return loss
