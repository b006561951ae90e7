"""MI355X-native efficient-probing (EP) head engine.

Drop-in for the attentive-pooling probe hot path of billpsomas/efficient-probing:
``poolings.ep.EfficientProbing``, ``probe_heads.build_probe_head`` / ``POOLINGS``,
``util.lars.LARS``, ``util.lr_sched.adjust_learning_rate`` and
``engine_finetune.train_one_epoch`` / ``evaluate`` keep the reference's names, arguments and
state-dict layout; the arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI
of ``include/ep_hip.h`` (``libep_hip.so``).
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401


def native_available() -> bool:
    import os
    return os.path.exists(_native.LIB_PATH)
