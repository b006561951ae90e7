"""Loss scaler and small distributed helpers with the reference's call surface
(reference util/misc.py:260-301, 397-405).  Only what the probe hot loop touches is here; the
logging / checkpoint helpers of the reference are not part of the accelerated path."""
from __future__ import annotations

import torch
import torch.distributed as dist

AMP_PRECISIONS = {
    "float16": torch.float16,
    "float32": torch.float32,
    "bfloat16": torch.bfloat16,
    "none": torch.float32,
}


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """Global gradient norm over ``parameters`` (reference util/misc.py:289-301)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    if float(norm_type) == float("inf"):
        return torch.stack([g.abs().max() for g in grads]).max()
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, norm_type) for g in grads]), norm_type)


class NativeScalerWithGradNormCount:
    """GradScaler semantics (init scale 65536, growth x2 every 2000 clean steps, back-off x0.5 on
    overflow, step skipped on overflow) driven by the native optimizers: the unscale and the
    inf/nan test run inside the optimizer kernel (``optimizer.step(inv_scale=...)``) instead of
    as separate passes over the gradients.  Call signature of reference util/misc.py:266-280."""
    state_dict_key = "amp_scaler"

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self._scale = float(init_scale) if enabled else 1.0
        self._growth_factor = growth_factor
        self._backoff_factor = backoff_factor
        self._growth_interval = growth_interval
        self._growth_tracker = 0
        self._enabled = enabled
        self._device_sync = None          # set by engine.attach_scaler: the state lives on the GPU while a fused engine drives it
        self._device_push = None          # ... and host-side changes (load_state_dict, a module-path step) go back up through this

    def get_scale(self) -> float:
        if self._device_sync is not None:
            self._device_sync()
        return self._scale

    def update(self, found_inf: bool) -> None:
        if not self._enabled:
            return
        if found_inf:
            self._scale *= self._backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self._growth_interval:
                self._scale *= self._growth_factor
                self._growth_tracker = 0

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        if self._device_sync is not None:
            self._device_sync()
        (loss * self._scale).backward(create_graph=create_graph)
        if not update_grad:
            return None
        native = hasattr(optimizer, "last_found_inf")
        if native and clip_grad is None:
            optimizer.step(inv_scale=1.0 / self._scale)
            norm = optimizer.last_grad_norm
            found = bool(optimizer.last_found_inf.item()) if optimizer.last_found_inf is not None else False
        else:
            params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            inv = 1.0 / self._scale
            torch._foreach_mul_([p.grad for p in params], inv)
            if clip_grad is not None:
                assert parameters is not None
                norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
            else:
                norm = get_grad_norm_(params)
            found = not bool(torch.isfinite(norm))
            if not found:
                optimizer.step()
        self.update(found)
        if self._device_push is not None:
            self._device_push()
        return norm

    def state_dict(self):
        if self._device_sync is not None:
            self._device_sync()
        return {"scale": self._scale, "growth_factor": self._growth_factor,
                "backoff_factor": self._backoff_factor, "growth_interval": self._growth_interval,
                "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, state_dict):
        self._scale = float(state_dict["scale"])
        self._growth_factor = state_dict.get("growth_factor", self._growth_factor)
        self._backoff_factor = state_dict.get("backoff_factor", self._backoff_factor)
        self._growth_interval = state_dict.get("growth_interval", self._growth_interval)
        self._growth_tracker = int(state_dict.get("_growth_tracker", 0))
        if self._device_push is not None:
            self._device_push()


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def all_reduce_mean(x):
    """Mean of a python scalar over ranks (reference util/misc.py:397-405)."""
    world = get_world_size()
    if world == 1:
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(float(x), device=dev)
    dist.all_reduce(t)
    return (t / world).item()
