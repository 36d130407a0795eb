"""ctypes binding of libplconvlstm.so (the C ABI declared in include/plconvlstm.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is
no fallback: if it is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PLC_LIB_PATH: load another build of the library instead, e.g. an older one for a same-box comparison)
LIB_PATH = os.environ.get("PLC_LIB_PATH") or os.path.join(_HERE, "libplconvlstm.so")

ABI_VERSION = 102      # PLC_VERSION of the include/plconvlstm.h these bindings mirror
EXPORTS = (
    "plc_version", "plc_last_error",
    "plc_nchw_f32_to_cb8_bf16", "plc_cb8_bf16_to_nchw_f32", "plc_nchw_f32_to_cb8_f32", "plc_cb8_f32_to_nchw_f32",
    "plc_packed_weight_bytes", "plc_packed_bias_bytes", "plc_pack_conv_weight",
    "plc_convlstm_cell_fwd_cb8", "plc_convlstm_cell_workspace_bytes", "plc_convlstm_cell_fwd",
    "plc_convlstm_cell_noalloc_max_bytes", "plc_convlstm_cell_fwd_noalloc",
    "plc_fp8_packed_weight_bytes", "plc_pack_lstm_weight_fp8", "plc_cb8_bf16_to_cb16_fp8", "plc_convlstm_cell_fwd_fp8",
    "plc_generator_workspace_bytes", "plc_generator_rollout_fwd", "plc_generator_workspace_offset",
    "plc_convlstm_cell_bwd_workspace_bytes", "plc_convlstm_cell_bwd", "plc_lstm_pointwise_bwd", "plc_lstm_pointwise_bwd_bf16dc",
    "plc_lstm_pointwise_bwd_pair_bf16dc",
    "plc_generator_bwd_workspace_bytes", "plc_generator_rollout_bwd",
    "plc_combined_loss_workspace_bytes", "plc_combined_loss",
    "plc_clip_adam_workspace_bytes", "plc_clip_adam_step", "plc_clip_adam_step_guarded",
    "plc_dp_pack_grads", "plc_dp_unpack_grads",
)

_f = C.c_void_p   # device pointers travel as plain addresses


class RolloutDesc(C.Structure):
    """Mirror of ``plc_rollout_desc``."""
    _fields_ = [
        ("B", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("C1", C.c_int), ("C2", C.c_int), ("L", C.c_int),
        ("dem_H", C.c_int), ("dem_W", C.c_int), ("lu_H", C.c_int), ("lu_W", C.c_int),
        ("n_blocks", C.c_int), ("mid_H", C.c_int), ("mid_W", C.c_int), ("out_H", C.c_int), ("out_W", C.c_int),
        ("mid_scale_h", C.c_float), ("mid_scale_w", C.c_float),
        ("out_scale_h", C.c_float), ("out_scale_w", C.c_float),
        ("save_for_backward", C.c_int), ("gate_fp8", C.c_int),
        ("rain", _f), ("dem", _f), ("lu", _f), ("row_tab", _f), ("col_tab", _f),
        ("init_w", _f), ("init_b", _f), ("cell1_w", _f), ("cell1_b", _f), ("cell2_w", _f), ("cell2_b", _f),
        ("dem0_w", _f), ("dem0_b", _f), ("dem2_w", _f), ("dem2_b", _f),
        ("lu0_w", _f), ("lu0_b", _f), ("lu2_w", _f), ("lu2_b", _f),
        ("up_w", _f * 8), ("up_b", _f * 8),
        ("post0_w", _f), ("post0_b", _f), ("post2_w", _f), ("post2_b", _f),
        ("out", _f), ("workspace", _f), ("workspace_bytes", C.c_size_t),
    ]


class RolloutGrads(C.Structure):
    """Mirror of ``plc_rollout_grads``."""
    _fields_ = [
        ("grad_out", _f),
        ("init_w", _f), ("init_b", _f), ("cell1_w", _f), ("cell1_b", _f), ("cell2_w", _f), ("cell2_b", _f),
        ("dem0_w", _f), ("dem0_b", _f), ("dem2_w", _f), ("dem2_b", _f),
        ("lu0_w", _f), ("lu0_b", _f), ("lu2_w", _f), ("lu2_b", _f),
        ("up_w", _f * 8), ("up_b", _f * 8),
        ("post0_w", _f), ("post0_b", _f), ("post2_w", _f), ("post2_b", _f),
        ("workspace", _f), ("workspace_bytes", C.c_size_t),
        ("d_rain", _f), ("d_dem", _f), ("d_lu", _f),
    ]


class LossDesc(C.Structure):
    """Mirror of ``plc_loss_desc``."""
    _fields_ = [
        ("B", C.c_int), ("T", C.c_int), ("out_H", C.c_int), ("out_W", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("N", C.c_int), ("values_per_batch", C.c_int), ("weight_mode", C.c_int),
        ("scale_factor", C.c_float),
        ("lambda_point", C.c_float), ("lambda_conserve", C.c_float), ("lambda_smooth", C.c_float),
        ("lambda_temporal", C.c_float),
        ("pred", _f), ("lr_input", _f), ("s_coords", _f), ("s_values", _f),
        ("losses", _f), ("grad_pred", _f), ("workspace", _f), ("workspace_bytes", C.c_size_t),
    ]


class OptTensor(C.Structure):
    """Mirror of ``plc_opt_tensor``."""
    _fields_ = [("param", _f), ("grad", _f), ("exp_avg", _f), ("exp_avg_sq", _f),
                ("numel", C.c_longlong), ("update", C.c_int), ("restore", _f)]


class PointwiseBwdArgs(C.Structure):
    """Mirror of ``plc_pointwise_bwd_args``."""
    _fields_ = [("dh0", _f), ("dh0_image_bytes", C.c_longlong), ("dh1", _f), ("dh1_image_bytes", C.c_longlong),
                ("dc_next", _f), ("gates", _f), ("c_prev", _f), ("dz", _f), ("dc", _f), ("Ch", C.c_int)]


_lib = None


def lib():
    """Load (once) and return the shared library with argument types set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    i, sz, p = C.c_int, C.c_size_t, C.c_void_p
    L.plc_version.restype = i
    if L.plc_version() != ABI_VERSION:
        # the structures below mirror ONE version of include/plconvlstm.h: a library built from another one would
        # read their fields at the wrong offsets
        raise RuntimeError(f"{LIB_PATH} is version {L.plc_version()} of the C ABI, this binding is version {ABI_VERSION}: "
                           "rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    L.plc_last_error.restype = C.c_char_p
    for name in ("plc_nchw_f32_to_cb8_bf16", "plc_cb8_bf16_to_nchw_f32", "plc_nchw_f32_to_cb8_f32",
                 "plc_cb8_f32_to_nchw_f32"):
        fn = getattr(L, name)
        fn.restype = i
        fn.argtypes = [p, p, i, i, i, i, p]
    L.plc_packed_weight_bytes.restype = sz
    L.plc_packed_weight_bytes.argtypes = [i, i, i, i]
    L.plc_packed_bias_bytes.restype = sz
    L.plc_packed_bias_bytes.argtypes = [i]
    L.plc_pack_conv_weight.restype = i
    L.plc_pack_conv_weight.argtypes = [p, p, p, p, i, i, i, i, i, p]
    L.plc_convlstm_cell_fwd_cb8.restype = i
    L.plc_convlstm_cell_fwd_cb8.argtypes = [p, p, p, p, p, p, p, p, i, i, i, i, i, p]
    L.plc_convlstm_cell_workspace_bytes.restype = sz
    L.plc_convlstm_cell_workspace_bytes.argtypes = [i, i, i, i, i]
    L.plc_convlstm_cell_fwd.restype = i
    L.plc_convlstm_cell_fwd.argtypes = [p, p, p, p, p, p, p, p, sz, i, i, i, i, i, p]
    L.plc_convlstm_cell_noalloc_max_bytes.restype = sz
    L.plc_convlstm_cell_noalloc_max_bytes.argtypes = []
    L.plc_convlstm_cell_fwd_noalloc.restype = i
    L.plc_convlstm_cell_fwd_noalloc.argtypes = [p, p, p, p, p, p, p, i, i, i, i, i, p]
    L.plc_fp8_packed_weight_bytes.restype = sz
    L.plc_fp8_packed_weight_bytes.argtypes = [i, i]
    L.plc_pack_lstm_weight_fp8.restype = i
    L.plc_pack_lstm_weight_fp8.argtypes = [p, p, p, i, i, p]
    L.plc_cb8_bf16_to_cb16_fp8.restype = i
    L.plc_cb8_bf16_to_cb16_fp8.argtypes = [p, p, i, i, i, i, p]
    L.plc_convlstm_cell_fwd_fp8.restype = i
    L.plc_convlstm_cell_fwd_fp8.argtypes = [p] * 10 + [i, i, i, i, i, p]
    L.plc_generator_workspace_bytes.restype = sz
    L.plc_generator_workspace_bytes.argtypes = [C.POINTER(RolloutDesc)]
    L.plc_generator_rollout_fwd.restype = i
    L.plc_generator_rollout_fwd.argtypes = [C.POINTER(RolloutDesc), p]
    L.plc_generator_workspace_offset.restype = sz
    L.plc_generator_workspace_offset.argtypes = [C.POINTER(RolloutDesc), C.c_char_p]
    L.plc_convlstm_cell_bwd_workspace_bytes.restype = sz
    L.plc_convlstm_cell_bwd_workspace_bytes.argtypes = [i, i, i, i, i]
    L.plc_convlstm_cell_bwd.restype = i
    L.plc_convlstm_cell_bwd.argtypes = [p] * 13 + [sz, i, i, i, i, i, p]
    L.plc_lstm_pointwise_bwd.restype = i
    L.plc_lstm_pointwise_bwd.argtypes = [p, C.c_longlong, p, C.c_longlong, p, C.c_longlong, p, p, p, p, p, i, i, i, i, p]
    L.plc_lstm_pointwise_bwd_bf16dc.restype = i
    L.plc_lstm_pointwise_bwd_bf16dc.argtypes = [p, C.c_longlong, p, C.c_longlong, p, C.c_longlong, p, p, p, p, p, i, i, i, i, p]
    L.plc_lstm_pointwise_bwd_pair_bf16dc.restype = i
    L.plc_lstm_pointwise_bwd_pair_bf16dc.argtypes = [C.POINTER(PointwiseBwdArgs), C.POINTER(PointwiseBwdArgs), i, i, i, p]
    L.plc_generator_bwd_workspace_bytes.restype = sz
    L.plc_generator_bwd_workspace_bytes.argtypes = [C.POINTER(RolloutDesc)]
    L.plc_generator_rollout_bwd.restype = i
    L.plc_generator_rollout_bwd.argtypes = [C.POINTER(RolloutDesc), C.POINTER(RolloutGrads), p]
    L.plc_combined_loss_workspace_bytes.restype = sz
    L.plc_combined_loss_workspace_bytes.argtypes = [C.POINTER(LossDesc)]
    L.plc_combined_loss.restype = i
    L.plc_combined_loss.argtypes = [C.POINTER(LossDesc), p]
    L.plc_clip_adam_workspace_bytes.restype = sz
    L.plc_clip_adam_workspace_bytes.argtypes = [C.POINTER(OptTensor), i]
    L.plc_clip_adam_step.restype = i
    L.plc_clip_adam_step.argtypes = [C.POINTER(OptTensor), i, C.c_double, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_double, C.c_longlong, p, p, sz, p]
    L.plc_clip_adam_step_guarded.restype = i
    L.plc_clip_adam_step_guarded.argtypes = [C.POINTER(OptTensor), i, C.c_double, C.c_double, C.c_double, C.c_double,
                                             C.c_double, C.c_double, p, p, p, p, p, sz, p]
    L.plc_dp_pack_grads.restype = i
    L.plc_dp_pack_grads.argtypes = [C.POINTER(OptTensor), i, p, C.c_longlong, C.c_float, p, p]
    L.plc_dp_unpack_grads.restype = i
    L.plc_dp_unpack_grads.argtypes = [C.POINTER(OptTensor), i, p, C.c_longlong, p, p, p]
    _lib = L
    return L


def check(rc: int, what: str):
    """Raise with the library's message if a call returned an error code."""
    if rc != 0:
        msg = lib().plc_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
