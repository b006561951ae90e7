"""ctypes binding of ``libep_hip.so`` (C ABI declared in ``include/ep_hip.h``).

The product path has no CPU fallback: if the shared library is missing, or a kernel is asked
to run without a GPU, this module raises -- loudly -- instead of silently computing the
result some other way.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EP_HIP_LIB") or os.path.join(_HERE, "libep_hip.so")   # EP_HIP_LIB: A/B builds only

EP_ABI_VERSION = 26        # include/ep_hip.h EP_ABI_VERSION the ctypes structs below are written for (checked in load())
EP_DTYPE_F32 = 0
EP_DTYPE_BF16 = 1
EP_DTYPE_F16 = 2          # fp16-stored tokens: forward entry points of the EP head only (ABI v24)
EP_ARITH_F32, EP_ARITH_BF16_AUTOCAST = 0, 1          # ep_head_step.arith (ABI v25)

c_f32p = C.c_void_p       # device pointers travel as integers (tensor.data_ptr())
c_i64 = C.c_int64
c_int = C.c_int
c_float = C.c_float
c_size = C.c_size_t
c_void = C.c_void_p


class EPSegment(C.Structure):
    _fields_ = [("offset", C.c_int64), ("numel", C.c_int64), ("apply_trust", C.c_int32),
                ("reserved", C.c_int32)]


class EPHeadDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("Q", C.c_int32),
                ("d_out", C.c_int32), ("C", C.c_int32)]


class EPHeadStep(C.Structure):
    _fields_ = [
        ("dims", EPHeadDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("image_index", C.c_void_p),
        ("targets", C.c_void_p),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("opt_state0", C.c_void_p), ("opt_state1", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("stats", C.c_void_p),
        ("found_inf", C.c_void_p), ("grad_norm", C.c_void_p),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
        ("grad_scale", C.c_float), ("inv_scale", C.c_float),
        ("accumulate", C.c_int32), ("optimizer", C.c_int32),
        ("lr", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float),
        ("trust_coefficient", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("opt_step", C.c_int64),
        ("phases", C.c_int32),
        ("aux_stream", C.c_void_p),
        ("opt_first_segment", C.c_int32), ("opt_num_segments", C.c_int32),
        ("defer_event", C.c_void_p),
        ("planes_valid", C.c_int32),
        ("arith", C.c_int32),
        ("scaler_state", C.c_void_p),
        ("scaler_slot", C.c_int32),
        ("scaler_growth", C.c_float), ("scaler_backoff", C.c_float),
        ("scaler_interval", C.c_int32),
    ]


class EPCocaDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("dh", C.c_int32),
                ("M", C.c_int32), ("C", C.c_int32)]


class EPCocaParams(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("img_queries", C.c_void_p), ("to_q", C.c_void_p),
                ("to_kv", C.c_void_p), ("to_out", C.c_void_p)]


class EPCocaStep(C.Structure):
    _fields_ = [
        ("dims", EPCocaDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("image_index", C.c_void_p),
        ("targets", C.c_void_p),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("opt_state0", C.c_void_p), ("opt_state1", C.c_void_p),
        ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("stats", C.c_void_p),
        ("found_inf", C.c_void_p), ("grad_norm", C.c_void_p),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
        ("grad_scale", C.c_float), ("inv_scale", C.c_float),
        ("accumulate", C.c_int32), ("optimizer", C.c_int32),
        ("lr", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float),
        ("trust_coefficient", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("opt_step", C.c_int64),
        ("phases", C.c_int32),
        ("aux_stream", C.c_void_p),
        ("arith", C.c_int32),
    ]


class EPAbmilpDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("C", C.c_int32)]


class EPAbmilpParams(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("w1", C.c_void_p),
                ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p)]


class EPAbmilpStep(C.Structure):
    _fields_ = [
        ("dims", EPAbmilpDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("targets", C.c_void_p),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("opt_state0", C.c_void_p), ("opt_state1", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("stats", C.c_void_p),
        ("found_inf", C.c_void_p), ("grad_norm", C.c_void_p),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
        ("grad_scale", C.c_float), ("inv_scale", C.c_float),
        ("accumulate", C.c_int32), ("optimizer", C.c_int32),
        ("lr", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float),
        ("trust_coefficient", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("opt_step", C.c_int64),
        ("phases", C.c_int32),
        ("arith", C.c_int32),
    ]


class EPSiglipDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("hidden", C.c_int32),
                ("C", C.c_int32)]


class EPSiglipParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("latent", "q_w", "q_b", "kv_w", "kv_b", "proj_w", "proj_b", "fc1_w", "fc1_b",
                                          "fc2_w", "fc2_b")]


class EPSiglipStep(C.Structure):
    _fields_ = [
        ("dims", EPSiglipDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("image_index", C.c_void_p),
        ("targets", C.c_void_p),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("opt_state0", C.c_void_p), ("opt_state1", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("stats", C.c_void_p),
        ("found_inf", C.c_void_p), ("grad_norm", C.c_void_p),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
        ("grad_scale", C.c_float), ("inv_scale", C.c_float),
        ("accumulate", C.c_int32), ("optimizer", C.c_int32),
        ("lr", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float),
        ("trust_coefficient", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("opt_step", C.c_int64),
        ("phases", C.c_int32),
        ("aux_stream", C.c_void_p),
    ]


class EPCaeDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("C", C.c_int32)]


class EPCaeParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("query", "nq_w", "nq_b", "nk_w", "nk_b", "nv_w", "nv_b", "n2_w", "n2_b", "q_w",
                                          "k_w", "v_w", "proj_w", "proj_b")]


class EPCaeStep(C.Structure):
    _fields_ = [
        ("dims", EPCaeDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("image_index", C.c_void_p),
        ("token_stats", C.c_void_p), ("ln_eps", C.c_float),
        ("targets", C.c_void_p),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("opt_state0", C.c_void_p), ("opt_state1", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("stats", C.c_void_p),
        ("found_inf", C.c_void_p), ("grad_norm", C.c_void_p),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
        ("grad_scale", C.c_float), ("inv_scale", C.c_float),
        ("accumulate", C.c_int32), ("optimizer", C.c_int32),
        ("lr", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float),
        ("trust_coefficient", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("opt_step", C.c_int64),
        ("phases", C.c_int32),
        ("aux_stream", C.c_void_p),
    ]


class EPJepaDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("hidden", C.c_int32),
                ("C", C.c_int32)]


class EPJepaParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("query", "n1_w", "n1_b", "q_w", "q_b", "kv_w", "kv_b", "proj_w", "proj_b", "n2_w",
                                          "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class EPJepaStep(C.Structure):
    _fields_ = [("dims", EPJepaDims)] + list(EPCaeStep._fields_[1:])


class EPAimDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("C", C.c_int32)]


class EPAimParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cls_token", "k_w", "v_w")]


class EPAimStep(C.Structure):
    _fields_ = [
        ("dims", EPAimDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("image_index", C.c_void_p),
        ("image_stats", C.c_void_p),
        ("tok_running_mean", C.c_void_p), ("tok_running_var", C.c_void_p), ("tok_num_batches_tracked", C.c_void_p),
        ("tok_bn_eps", C.c_float), ("tok_bn_momentum", C.c_float),
    ] + list(EPCaeStep._fields_[7:])                 # targets ... aux_stream, as in every head step


class EPSimpoolDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("C", C.c_int32),
                ("linears", C.c_int32)]


class EPSimpoolParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_w", "norm_b", "wq", "wk")]


class EPSimpoolStep(C.Structure):
    _fields_ = [
        ("dims", EPSimpoolDims),
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("x_bstride", C.c_int64),
        ("image_index", C.c_void_p),
        ("token_stats", C.c_void_p), ("image_stats", C.c_void_p), ("ln_eps", C.c_float),
    ] + list(EPCaeStep._fields_[7:])                 # targets ... aux_stream, as in every head step


class EPCaitDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("hidden", C.c_int32),
                ("C", C.c_int32), ("ln_eps", C.c_float), ("final_eps", C.c_float)]


class EPCaitParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cls_token", "gamma_1", "gamma_2", "n1_w", "n1_b", "q_w", "q_b", "k_w", "k_b", "v_w",
                                          "v_b", "proj_w", "proj_b", "n2_w", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "norm_w",
                                          "norm_b")]


class EPCaitStep(C.Structure):
    _fields_ = [("dims", EPCaitDims)] + list(EPCaeStep._fields_[1:])


class EPClipDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("ln_eps", C.c_float)]


class EPClipParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos_embed", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm_w", "norm_b")]


class EPClipStep(C.Structure):
    _fields_ = [("dims", EPClipDims)] + list(EPCaeStep._fields_[1:]) + [("xhat_mean", C.c_void_p)]


class EPDolgDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("C", C.c_int32)]


class EPDolgParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("conv1_w", "conv1_b", "bn_w", "bn_b", "conv2_w", "conv2_b")]


class EPDolgStep(C.Structure):
    _fields_ = [("dims", EPDolgDims)] + list(EPAimStep._fields_[1:])


class EPCbamDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("C", C.c_int32), ("rd", C.c_int32), ("ks", C.c_int32)]


class EPCbamParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fc1_w", "fc2_w", "conv_w", "bn_w", "bn_b")]


class EPCbamStep(C.Structure):
    _fields_ = [("dims", EPCbamDims)] + list(EPAimStep._fields_[1:])


class EPDinovitDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("hidden", C.c_int32), ("C", C.c_int32),
                ("ln_eps", C.c_float)]


class EPDinovitParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("n1_w", "n1_b", "qkv_w", "proj_w", "proj_b", "n2_w", "n2_b", "fc1_w", "fc1_b", "fc2_w",
                                          "fc2_b")]


class EPDinovitStep(C.Structure):
    _fields_ = [("dims", EPDinovitDims)] + list(EPAbmilpStep._fields_[1:])


# name -> (restype, argtypes); every symbol include/ep_hip.h declares
SIGNATURES = {
    "ep_version": (c_int, []),
    "ep_last_error_string": (C.c_char_p, []),
    "ep_device_cu_count": (c_int, []),
    "ep_debug_force_generic_pool": (c_int, [c_int]),
    "ep_pool_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "ep_pool_kernel_name": (C.c_char_p, [c_int, c_int, c_int, c_int, c_int]),
    "ep_token_stats": (c_int, [c_void, c_int, c_i64, c_int, c_int, c_int, c_float, c_f32p, c_void]),
    "ep_pool_forward_ln": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_f32p, c_i64, c_int, c_float, c_f32p,
                                   c_f32p, c_f32p, c_f32p, c_void, c_size, c_void]),
    "ep_pool_backward_ln": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_float, c_f32p, c_f32p, c_f32p,
                                    c_f32p, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_pool_kernel_name_ex": (C.c_char_p, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ep_linear_kernel_name": (C.c_char_p, [c_int, c_int, c_int]),
    "ep_pool_forward": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_f32p, c_i64, c_int, c_float,
                                c_f32p, c_f32p, c_f32p, c_void, c_size, c_void]),
    "ep_pool_backward": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_float, c_f32p, c_f32p,
                                 c_f32p, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_pool_backward_per_image": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_float, c_f32p, c_f32p,
                                           c_f32p, c_f32p, c_void]),
    "ep_attention_from_scores": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, c_void]),
    "ep_project_forward": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_f32p, c_void]),
    "ep_project_backward": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_f32p,
                                    c_f32p, c_f32p, c_int, c_void]),
    "ep_bn_workspace_bytes": (c_size, [c_int, c_int]),
    "ep_bn_forward_train": (c_int, [c_f32p, c_int, c_int, c_float, c_float, c_f32p, c_f32p, c_f32p, c_f32p,
                                    c_void, c_void, c_size, c_void]),
    "ep_bn_forward_eval": (c_int, [c_f32p, c_int, c_int, c_float, c_f32p, c_f32p, c_f32p, c_void]),
    "ep_bn_backward": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_f32p, c_void, c_size, c_void]),
    "ep_linear_forward": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, c_int, c_void]),
    "ep_linear_backward": (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p,
                                   c_int, c_void]),
    "ep_planes_elems": (c_size, [c_int, c_int]),
    "ep_planes_split": (c_int, [c_f32p, c_int, c_int, c_i64, c_void, c_void, c_void]),
    "ep_matmul_planes": (c_int, [c_f32p, c_i64, c_void, c_int, c_int, c_f32p, c_int, c_int, c_f32p, c_i64, c_void]),
    "ep_cross_entropy": (c_int, [c_f32p, c_int, c_void, c_int, c_int, c_float, c_f32p, c_f32p, c_f32p, c_void]),
    "ep_optim_workspace_bytes": (c_size, [c_i64, c_int]),
    "ep_lars_step": (c_int, [c_f32p, c_f32p, c_f32p, c_i64, C.POINTER(EPSegment), c_int, c_float, c_float,
                             c_float, c_float, c_float, c_void, c_f32p, c_void, c_size, c_void]),
    "ep_sgd_step": (c_int, [c_f32p, c_f32p, c_i64, c_float, c_float, c_float, c_void, c_f32p, c_void, c_size,
                            c_void]),
    "ep_adamw_step": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_float, c_float, c_float, c_float,
                              c_float, c_float, c_void, c_f32p, c_void, c_size, c_void]),
    "ep_head_param_offsets": (c_i64, [C.POINTER(EPHeadDims), C.POINTER(c_i64)]),
    "ep_head_workspace_bytes": (c_size, [C.POINTER(EPHeadDims)]),
    "ep_head_workspace_flag_offset": (C.c_int64, [C.POINTER(EPHeadDims)]),
    "ep_head_workspace_logits_offset": (C.c_int64, [C.POINTER(EPHeadDims), C.POINTER(C.c_int32)]),
    "ep_head_workspace_init": (c_int, [C.POINTER(EPHeadDims), c_void, c_size, c_void]),
    "ep_debug_set_pass_events": (c_int, [c_void, c_void, c_void, c_void]),
    "ep_head_train_step": (c_int, [C.POINTER(EPHeadStep), c_void, c_size, c_void]),
    "ep_head_eval_forward": (c_int, [C.POINTER(EPHeadDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_f32p,
                                     c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_lp_param_offsets": (c_i64, [C.POINTER(EPHeadDims), C.POINTER(c_i64)]),
    "ep_lp_workspace_bytes": (c_size, [C.POINTER(EPHeadDims)]),
    "ep_lp_train_step": (c_int, [C.POINTER(EPHeadStep), c_void, c_size, c_void]),
    "ep_lp_eval_forward": (c_int, [C.POINTER(EPHeadDims), c_f32p, c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_void,
                                   c_size, c_void]),
    "ep_coca_pool_workspace_bytes": (c_size, [C.POINTER(EPCocaDims)]),
    "ep_coca_pool_forward": (c_int, [C.POINTER(EPCocaDims), c_void, c_int, c_i64, c_void, C.POINTER(EPCocaParams),
                                     c_float, c_f32p, c_void, c_size, c_void]),
    "ep_coca_pool_backward": (c_int, [C.POINTER(EPCocaDims), c_void, c_int, c_i64, c_void, C.POINTER(EPCocaParams),
                                      c_f32p, C.POINTER(EPCocaParams), c_int, c_void, c_size, c_void]),
    "ep_coca_attention": (c_int, [C.POINTER(EPCocaDims), c_void, c_f32p, c_void]),
    "ep_coca_head_param_offsets": (c_i64, [C.POINTER(EPCocaDims), C.POINTER(c_i64)]),
    "ep_coca_head_workspace_bytes": (c_size, [C.POINTER(EPCocaDims)]),
    "ep_coca_head_workspace_logits_offset": (C.c_int64, [C.POINTER(EPCocaDims), C.POINTER(C.c_int32)]),
    "ep_coca_head_train_step": (c_int, [C.POINTER(EPCocaStep), c_void, c_size, c_void]),
    "ep_abmilp_pool_workspace_bytes": (c_size, [C.POINTER(EPAbmilpDims)]),
    "ep_abmilp_pool_forward": (c_int, [C.POINTER(EPAbmilpDims), c_void, c_int, c_i64, C.POINTER(EPAbmilpParams),
                                       c_f32p, c_f32p, c_void, c_size, c_void]),
    "ep_abmilp_pool_backward": (c_int, [C.POINTER(EPAbmilpDims), c_void, c_int, c_i64, C.POINTER(EPAbmilpParams),
                                        c_f32p, C.POINTER(EPAbmilpParams), c_int, c_void, c_size, c_void]),
    "ep_abmilp_head_param_offsets": (c_i64, [C.POINTER(EPAbmilpDims), C.POINTER(c_i64)]),
    "ep_abmilp_head_workspace_bytes": (c_size, [C.POINTER(EPAbmilpDims)]),
    "ep_abmilp_head_workspace_logits_offset": (C.c_int64, [C.POINTER(EPAbmilpDims), C.POINTER(C.c_int32)]),
    "ep_abmilp_head_train_step": (c_int, [C.POINTER(EPAbmilpStep), c_void, c_size, c_void]),
    "ep_abmilp_head_eval_forward": (c_int, [C.POINTER(EPAbmilpDims), c_void, c_int, c_i64, c_f32p, c_f32p, c_f32p,
                                            c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_l2_normalize": (c_int, [c_f32p, c_i64, c_int, c_float, c_f32p, c_void]),
    "ep_knn_workspace_bytes": (c_size, [c_int, c_int]),
    "ep_knn_workspace_bytes_ex": (c_size, [c_int, c_int, c_int]),
    "ep_knn_topk": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_f32p, c_void, c_int, c_void, c_size, c_void]),
    "ep_knn_vote": (c_int, [c_f32p, c_void, c_int, c_void, c_int, c_int, c_float, c_int, c_void, c_void, c_f32p, c_void]),
    "ep_siglip_pool_workspace_bytes": (c_size, [C.POINTER(EPSiglipDims)]),
    "ep_siglip_pool_forward": (c_int, [C.POINTER(EPSiglipDims), c_void, c_int, c_i64, c_void, C.POINTER(EPSiglipParams),
                                       c_f32p, c_void, c_size, c_void]),
    "ep_siglip_pool_backward": (c_int, [C.POINTER(EPSiglipDims), c_void, c_int, c_i64, c_void, C.POINTER(EPSiglipParams),
                                        c_f32p, C.POINTER(EPSiglipParams), c_int, c_void, c_size, c_void]),
    "ep_siglip_attention": (c_int, [C.POINTER(EPSiglipDims), c_void, c_f32p, c_void]),
    "ep_siglip_head_param_offsets": (c_i64, [C.POINTER(EPSiglipDims), C.POINTER(c_i64)]),
    "ep_siglip_head_workspace_bytes": (c_size, [C.POINTER(EPSiglipDims)]),
    "ep_siglip_head_train_step": (c_int, [C.POINTER(EPSiglipStep), c_void, c_size, c_void]),
    "ep_siglip_head_eval_forward": (c_int, [C.POINTER(EPSiglipDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_f32p,
                                            c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_cae_pool_workspace_bytes": (c_size, [C.POINTER(EPCaeDims)]),
    "ep_cae_pool_forward": (c_int, [C.POINTER(EPCaeDims), c_void, c_int, c_i64, c_void, c_f32p, c_float, C.POINTER(EPCaeParams),
                                    c_f32p, c_void, c_size, c_void]),
    "ep_cae_pool_backward": (c_int, [C.POINTER(EPCaeDims), c_void, c_int, c_i64, c_void, c_f32p, c_float, C.POINTER(EPCaeParams),
                                     c_f32p, C.POINTER(EPCaeParams), c_int, c_void, c_size, c_void]),
    "ep_cae_head_param_offsets": (c_i64, [C.POINTER(EPCaeDims), C.POINTER(c_i64)]),
    "ep_cae_head_workspace_bytes": (c_size, [C.POINTER(EPCaeDims)]),
    "ep_cae_head_train_step": (c_int, [C.POINTER(EPCaeStep), c_void, c_size, c_void]),
    "ep_cae_head_eval_forward": (c_int, [C.POINTER(EPCaeDims), c_void, c_int, c_i64, c_void, c_f32p, c_float, c_f32p, c_f32p,
                                         c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_jepa_pool_workspace_bytes": (c_size, [C.POINTER(EPJepaDims)]),
    "ep_jepa_pool_forward": (c_int, [C.POINTER(EPJepaDims), c_void, c_int, c_i64, c_void, c_f32p, c_float, C.POINTER(EPJepaParams),
                                     c_f32p, c_void, c_size, c_void]),
    "ep_jepa_pool_backward": (c_int, [C.POINTER(EPJepaDims), c_void, c_int, c_i64, c_void, c_f32p, C.POINTER(EPJepaParams),
                                      c_f32p, C.POINTER(EPJepaParams), c_int, c_void, c_size, c_void]),
    "ep_jepa_head_param_offsets": (c_i64, [C.POINTER(EPJepaDims), C.POINTER(c_i64)]),
    "ep_jepa_head_workspace_bytes": (c_size, [C.POINTER(EPJepaDims)]),
    "ep_jepa_head_train_step": (c_int, [C.POINTER(EPJepaStep), c_void, c_size, c_void]),
    "ep_jepa_head_eval_forward": (c_int, [C.POINTER(EPJepaDims), c_void, c_int, c_i64, c_void, c_f32p, c_float, c_f32p, c_f32p,
                                          c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_coca_head_eval_forward": (c_int, [C.POINTER(EPCocaDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p,
                                          c_float, c_f32p, c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_imgq_pool_forward": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_int, c_f32p,
                                     c_f32p, c_void]),
    "ep_imgq_pool_backward": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_int, c_f32p,
                                      c_f32p, c_f32p, c_f32p, c_void]),
    "ep_simpool_pool_workspace_bytes": (c_size, [C.POINTER(EPSimpoolDims)]),
    "ep_simpool_pool_forward": (c_int, [C.POINTER(EPSimpoolDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_float,
                                        C.POINTER(EPSimpoolParams), c_f32p, c_void, c_size, c_void]),
    "ep_simpool_pool_backward": (c_int, [C.POINTER(EPSimpoolDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_float,
                                         C.POINTER(EPSimpoolParams), c_f32p, c_f32p, C.POINTER(EPSimpoolParams), c_int, c_void,
                                         c_size, c_void]),
    "ep_simpool_attention": (c_int, [C.POINTER(EPSimpoolDims), c_void, c_int, c_i64, c_void, c_f32p, c_void, c_f32p, c_void]),
    "ep_simpool_head_param_offsets": (c_i64, [C.POINTER(EPSimpoolDims), C.POINTER(c_i64)]),
    "ep_simpool_head_workspace_bytes": (c_size, [C.POINTER(EPSimpoolDims)]),
    "ep_simpool_head_train_step": (c_int, [C.POINTER(EPSimpoolStep), c_void, c_size, c_void]),
    "ep_simpool_head_eval_forward": (c_int, [C.POINTER(EPSimpoolDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_float,
                                             c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_cait_pool_workspace_bytes": (c_size, [C.POINTER(EPCaitDims)]),
    "ep_cait_pool_forward": (c_int, [C.POINTER(EPCaitDims), c_void, c_int, c_i64, c_void, c_f32p, C.POINTER(EPCaitParams), c_f32p,
                                     c_void, c_size, c_void]),
    "ep_cait_pool_backward": (c_int, [C.POINTER(EPCaitDims), c_void, c_int, c_i64, c_void, c_f32p, C.POINTER(EPCaitParams), c_f32p,
                                      C.POINTER(EPCaitParams), c_int, c_void, c_size, c_void]),
    "ep_cait_head_param_offsets": (c_i64, [C.POINTER(EPCaitDims), C.POINTER(c_i64)]),
    "ep_cait_head_workspace_bytes": (c_size, [C.POINTER(EPCaitDims)]),
    "ep_cait_head_train_step": (c_int, [C.POINTER(EPCaitStep), c_void, c_size, c_void]),
    "ep_cait_head_eval_forward": (c_int, [C.POINTER(EPCaitDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_f32p, c_f32p,
                                          c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_token_xhat_mean": (c_int, [c_void, c_int, c_i64, c_void, c_f32p, c_int, c_int, c_int, c_f32p, c_void]),
    "ep_clip_pool_workspace_bytes": (c_size, [C.POINTER(EPClipDims)]),
    "ep_clip_pool_forward": (c_int, [C.POINTER(EPClipDims), c_void, c_int, c_i64, c_void, c_f32p, C.POINTER(EPClipParams), c_f32p,
                                     c_void, c_size, c_void]),
    "ep_clip_pool_backward": (c_int, [C.POINTER(EPClipDims), c_void, c_int, c_i64, c_void, c_f32p, C.POINTER(EPClipParams), c_f32p,
                                      C.POINTER(EPClipParams), c_int, c_void, c_size, c_void]),
    "ep_clip_attention": (c_int, [C.POINTER(EPClipDims), c_void, c_f32p, c_void]),
    "ep_clip_head_param_offsets": (c_i64, [C.POINTER(EPClipDims), C.POINTER(c_i64)]),
    "ep_clip_head_workspace_bytes": (c_size, [C.POINTER(EPClipDims)]),
    "ep_clip_head_train_step": (c_int, [C.POINTER(EPClipStep), c_void, c_size, c_void]),
    "ep_clip_head_eval_forward": (c_int, [C.POINTER(EPClipDims), c_void, c_int, c_i64, c_void, c_f32p, c_f32p, c_f32p, c_f32p,
                                          c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_dolg_pool_workspace_bytes": (c_size, [C.POINTER(EPDolgDims)]),
    "ep_dolg_pool_forward": (c_int, [C.POINTER(EPDolgDims), c_void, c_int, c_i64, c_int, c_float, c_float, c_f32p, c_f32p, c_void,
                                     C.POINTER(EPDolgParams), c_f32p, c_void, c_size, c_void]),
    "ep_dolg_pool_backward": (c_int, [C.POINTER(EPDolgDims), c_void, c_int, c_i64, C.POINTER(EPDolgParams), c_f32p,
                                      C.POINTER(EPDolgParams), c_int, c_void, c_size, c_void]),
    "ep_dolg_attention": (c_int, [C.POINTER(EPDolgDims), c_void, c_f32p, c_void]),
    "ep_dolg_head_param_offsets": (c_i64, [C.POINTER(EPDolgDims), C.POINTER(c_i64)]),
    "ep_dolg_head_workspace_bytes": (c_size, [C.POINTER(EPDolgDims)]),
    "ep_dolg_head_train_step": (c_int, [C.POINTER(EPDolgStep), c_void, c_size, c_void]),
    "ep_dolg_head_eval_forward": (c_int, [C.POINTER(EPDolgDims), c_void, c_int, c_i64, c_float, c_f32p, c_f32p, c_f32p, c_f32p,
                                          c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_cbam_channel_table": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_f32p, c_void]),
    "ep_cbam_pool_workspace_bytes": (c_size, [C.POINTER(EPCbamDims)]),
    "ep_cbam_pool_forward": (c_int, [C.POINTER(EPCbamDims), c_void, c_int, c_i64, c_void, c_f32p, c_int, c_float, c_float, c_f32p,
                                     c_f32p, c_void, C.POINTER(EPCbamParams), c_f32p, c_void, c_size, c_void]),
    "ep_cbam_pool_backward": (c_int, [C.POINTER(EPCbamDims), c_void, c_int, c_i64, c_void, C.POINTER(EPCbamParams), c_f32p,
                                      C.POINTER(EPCbamParams), c_int, c_void, c_size, c_void]),
    "ep_cbam_head_param_offsets": (c_i64, [C.POINTER(EPCbamDims), C.POINTER(c_i64)]),
    "ep_cbam_head_workspace_bytes": (c_size, [C.POINTER(EPCbamDims)]),
    "ep_cbam_head_train_step": (c_int, [C.POINTER(EPCbamStep), c_void, c_size, c_void]),
    "ep_cbam_head_eval_forward": (c_int, [C.POINTER(EPCbamDims), c_void, c_int, c_i64, c_void, c_f32p, c_float, c_f32p, c_f32p,
                                          c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
    "ep_dinovit_pool_workspace_bytes": (c_size, [C.POINTER(EPDinovitDims)]),
    "ep_dinovit_pool_forward": (c_int, [C.POINTER(EPDinovitDims), c_void, c_int, c_i64, C.POINTER(EPDinovitParams), c_f32p, c_void,
                                        c_size, c_void]),
    "ep_dinovit_pool_backward": (c_int, [C.POINTER(EPDinovitDims), c_void, c_int, c_i64, C.POINTER(EPDinovitParams), c_f32p,
                                         C.POINTER(EPDinovitParams), c_int, c_void, c_size, c_void]),
    "ep_dinovit_attention": (c_int, [C.POINTER(EPDinovitDims), c_void, c_f32p, c_void]),
    "ep_dinovit_head_param_offsets": (c_i64, [C.POINTER(EPDinovitDims), C.POINTER(c_i64)]),
    "ep_dinovit_head_workspace_bytes": (c_size, [C.POINTER(EPDinovitDims)]),
    "ep_dinovit_head_train_step": (c_int, [C.POINTER(EPDinovitStep), c_void, c_size, c_void]),
    "ep_dinovit_head_eval_forward": (c_int, [C.POINTER(EPDinovitDims), c_void, c_int, c_i64, c_f32p, c_f32p, c_f32p, c_float, c_f32p,
                                             c_int, c_void, c_size, c_void]),
    "ep_rowq_pool_forward": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                     c_f32p, c_void]),
    "ep_rowq_pool_backward": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                      c_f32p, c_f32p, c_void]),
    "ep_channel_stats": (c_int, [c_void, c_int, c_i64, c_void, c_int, c_int, c_int, c_f32p, c_void]),
    "ep_aim_pool_workspace_bytes": (c_size, [C.POINTER(EPAimDims)]),
    "ep_aim_pool_forward": (c_int, [C.POINTER(EPAimDims), c_void, c_int, c_i64, c_void, c_f32p, c_int, c_float, c_float, c_f32p,
                                    c_f32p, c_void, C.POINTER(EPAimParams), c_f32p, c_void, c_size, c_void]),
    "ep_aim_pool_backward": (c_int, [C.POINTER(EPAimDims), c_void, c_int, c_i64, c_void, C.POINTER(EPAimParams), c_f32p, c_f32p,
                                     C.POINTER(EPAimParams), c_int, c_void, c_size, c_void]),
    "ep_aim_attention": (c_int, [C.POINTER(EPAimDims), c_void, c_f32p, c_void]),
    "ep_aim_head_param_offsets": (c_i64, [C.POINTER(EPAimDims), C.POINTER(c_i64)]),
    "ep_aim_head_workspace_bytes": (c_size, [C.POINTER(EPAimDims)]),
    "ep_aim_head_train_step": (c_int, [C.POINTER(EPAimStep), c_void, c_size, c_void]),
    "ep_aim_head_eval_forward": (c_int, [C.POINTER(EPAimDims), c_void, c_int, c_i64, c_void, c_float, c_f32p, c_f32p, c_f32p,
                                         c_f32p, c_f32p, c_float, c_f32p, c_int, c_void, c_size, c_void]),
}

_lib: Optional[C.CDLL] = None


class NativeLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the shared library.  Raises NativeLibraryError when it is absent:
    build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
    ``efficient_probing_amd/csrc/build.sh``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built (run efficient_probing_amd/csrc/build.sh). "
            "There is no CPU fallback for the EP head kernels.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    # the step structs above are written for ONE ABI version (they have grown across versions): a stale, git-ignored
    # .so that still exports every symbol would misread them silently
    got = int(lib.ep_version())
    if got != EP_ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH} reports ABI version {got}, the Python bindings are written for "
                                 f"{EP_ABI_VERSION}: rebuild it (efficient_probing_amd/csrc/build.sh)")
    _lib = lib
    return lib


def last_error() -> str:
    return load().ep_last_error_string().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = "invalid argument" if rc < 0 else "HIP error"
        raise RuntimeError(f"{what} failed ({kind} {rc}): {last_error()}")


def require_gpu_tensor(t, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the EP head kernels have no CPU path "
                           f"(got device {t.device})")


def current_stream_ptr(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
