"""ctypes binding of libtbg_hip.so (include/tbg.h).

Plumbing only: PyTorch provides device memory (caching allocator = the role of TF's
``allocate_output``) and the current HIP stream; every compute call goes through the C ABI.
There is NO fallback: if the library is missing or a tensor is not a contiguous fp32 device
tensor, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtbg_hip.so")

ACT_LINEAR, ACT_LRELU = 0, 1
SQRT2 = 1.4142135623730951

EXPORTS = [
    "tbg_version", "tbg_strerror", "tbg_crc32c", "tbg_upfirdn2d_f32", "tbg_upfirdn2d_ex_f32", "tbg_upfirdn2d_sep_f32", "tbg_conv2d_f32",
    "tbg_conv2d_wgrad_f32", "tbg_conv2d_wgrad_ex_f32", "tbg_conv2d_wgrad_workspace_bytes", "tbg_weight_pack_f32", "tbg_weight_pack_floats", "tbg_conv2d_kernel_name", "tbg_conv2d_f32_variant", "tbg_conv2d_bf16_variant", "tbg_conv2d_wgrad_kernel_name", "tbg_weight_pack_bf16_bytes", "tbg_weight_pack_bf16", "tbg_weight_pack_multi", "tbg_modconv_bwd_smalls_f32", "tbg_torgb_bwd_smalls_f32", "tbg_minibatch_std_fwd_f32", "tbg_minibatch_std_bwd_f32", "tbg_dense_fwd_f32", "tbg_dense_bwd_f32", "tbg_dense_multi_fwd_f32", "tbg_dense_multi_bwd_f32", "tbg_conv2d_bf16", "tbg_conv2d_bf16_kernel_name", "tbg_conv2d_wgrad_bf16", "tbg_conv2d_wgrad_bf16_kernel_name", "tbg_lstm_step_fwd_f32", "tbg_lstm_step_bwd_f32", "tbg_attn_ctx_fwd_f32", "tbg_attn_ctx_bwd_f32", "tbg_bias_act_fwd_f32", "tbg_slab_epilogue_f32", "tbg_bias_act_bwd_chunks",
    "tbg_upfirdn2d_kernel_name", "tbg_upfirdn2d_f16", "tbg_weight_pack_x3_bytes", "tbg_weight_pack_x3", "tbg_conv2d_x3", "tbg_conv2d_x3_kernel_name", "tbg_conv2d_x3_variant", "tbg_conv2d_wgrad_x3", "tbg_conv2d_wgrad_x3_kernel_name", "tbg_conv2d_dot_slots", "tbg_conv2d_blocks", "tbg_units_bytes", "tbg_units_pack_f32",
    "tbg_conv2d_wgrad_units", "tbg_conv2d_wgrad_units_workspace_bytes", "tbg_conv2d_units", "tbg_conv2d_units_dot_slots", "tbg_conv2d_units_blocks", "tbg_conv2d_units_tile_channels", "tbg_bias_act_bwd_units", "tbg_bias_act_bwd_units_chunks",
    "tbg_units_s2_bytes", "tbg_units_pack_s2_f32", "tbg_upfirdn2d_units_s2_f32", "tbg_conv2d_units_s2_blocks", "tbg_conv2d_units_s2_tile_channels", "tbg_conv2d_units_s2_dot_slots", "tbg_conv2d_units_s2", "tbg_conv2d_wgrad_units_s2_workspace_bytes", "tbg_conv2d_wgrad_units_s2", "tbg_conv2d_units_t2_blocks", "tbg_conv2d_units_t2",
    "tbg_slab_epilogue_units_f32", "tbg_conv2d_units_small", "tbg_conv2d_units_small_blocks", "tbg_conv2d_units_small_dot_slots", "tbg_conv2d_units_small_tile_pixels", "tbg_lstm_fused_fwd_f32", "tbg_lstm_fused_bwd_f32", "tbg_lstm_cell_fused_fwd_f32", "tbg_rows_gemv_t_f32", "tbg_dec_sample_fwd_f32", "tbg_dec_sample_bwd_f32",
    "tbg_bias_act_bwd_f32", "tbg_axpby_planes_f32", "tbg_bias_act_bwd2_f32", "tbg_rgb_project_f32", "tbg_rgb_backproject_f32", "tbg_rgb_backproject_chunks", "tbg_adam_tf_f32", "tbg_ema_lerp_f32", "tbg_demod_coefs_f32",
]


class TbgError(RuntimeError):
    pass


class Epilogue(C.Structure):
    _fields_ = [("out_scale", C.c_void_p), ("bias", C.c_void_p), ("noise", C.c_void_p), ("strength", C.c_void_p),
                ("residual", C.c_void_p), ("dot_aux", C.c_void_p), ("dot_out", C.c_void_p), ("gate", C.c_void_p), ("alpha", C.c_float), ("bias_mul", C.c_float), ("slope", C.c_float),
                ("gain", C.c_float), ("res_scale", C.c_float), ("act", C.c_int), ("res_first", C.c_int),
                # unit sink (tbg.h): units(out * units_scale) written beside / instead of the fp32 output
                ("units_out", C.c_void_p), ("units_scale", C.c_void_p), ("units_planes", C.c_int)]


class PackItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("T", C.c_int), ("I", C.c_int), ("O", C.c_int),
                ("transpose", C.c_int), ("flip", C.c_int), ("bf16", C.c_int)]


class DenseItem(C.Structure):  # tbg_dense_item
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("dout", C.c_void_p), ("out", C.c_void_p),
                ("dx", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("N", C.c_int), ("ldx", C.c_int)]


DENSE_MAX_ITEMS = 24


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "C", "M", "Hin", "Win", "Hout", "Wout", "KH", "KW", "sy", "sx", "py", "px",
                                       "transposed", "flip", "ldw", "ksplit")]


class WgradDesc(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("B", "CS", "CL", "Hs", "Ws", "Hl", "Wl", "KH", "KW", "sy", "sx", "py", "px",
                                        "st_t", "st_l", "st_s")] + [("alpha", C.c_float)] +
                # rider: the layer's bias gradient summed by the filter gradient's reduce launch (tbg.h)
                [("bias_parts", C.c_void_p), ("bias_grad", C.c_void_p), ("bias_B", C.c_int), ("bias_nch", C.c_int)])


_lib = None


def lib():
    """Load the HIP library (fails loudly -- there is no CPU/eager fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TbgError(f"{LIB_PATH} not found: run `python -m textboxgan_amd.build` (hipcc, gfx950)")
        l = C.CDLL(LIB_PATH)
        l.tbg_strerror.restype = C.c_char_p
        l.tbg_strerror.argtypes = [C.c_int]
        l.tbg_version.restype = C.c_int
        l.tbg_bias_act_bwd_chunks.restype = C.c_int
        l.tbg_bias_act_bwd_chunks.argtypes = [C.c_int]
        vp, ci, cf, ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
        l.tbg_upfirdn2d_f32.argtypes = [vp, vp, vp] + [ci] * 14 + [vp]
        l.tbg_upfirdn2d_ex_f32.argtypes = [vp, vp, vp] + [ci] * 13 + [vp, ci, C.POINTER(Epilogue), vp]
        l.tbg_upfirdn2d_sep_f32.argtypes = [vp, vp, vp, vp] + [ci] * 13 + [vp, ci, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_wgrad_f32.argtypes = [C.POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, ll, vp]
        l.tbg_conv2d_wgrad_ex_f32.argtypes = [C.POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, vp, cf, vp, ll, vp]
        l.tbg_conv2d_wgrad_workspace_bytes.restype = ll
        l.tbg_conv2d_wgrad_workspace_bytes.argtypes = [C.POINTER(WgradDesc)]
        l.tbg_weight_pack_f32.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        l.tbg_weight_pack_floats.argtypes = [ci, ci, ci, ci]
        l.tbg_weight_pack_floats.restype = C.c_longlong
        l.tbg_lstm_step_fwd_f32.argtypes = [vp] * 6 + [ci] * 5 + [vp]
        l.tbg_lstm_fused_fwd_f32.argtypes = [vp] * 7 + [ci] * 5 + [vp]
        l.tbg_lstm_fused_bwd_f32.argtypes = [vp] * 8 + [ci] * 6 + [vp]
        l.tbg_lstm_cell_fused_fwd_f32.argtypes = [vp] * 6 + [ci] * 5 + [vp]
        l.tbg_rows_gemv_t_f32.argtypes = [vp] * 3 + [ci] * 3 + [vp]
        l.tbg_dec_sample_fwd_f32.argtypes = [vp] * 14 + [ci] * 6 + [vp]
        l.tbg_dec_sample_bwd_f32.argtypes = [vp] * 16 + [ci] * 5 + [vp]
        l.tbg_lstm_step_bwd_f32.argtypes = [vp] * 7 + [ci] * 6 + [vp]
        l.tbg_attn_ctx_fwd_f32.argtypes = [vp] * 6 + [ci] * 4 + [vp]
        l.tbg_attn_ctx_bwd_f32.argtypes = [vp] * 9 + [ci] * 4 + [vp]
        l.tbg_conv2d_kernel_name.argtypes = [C.POINTER(ConvDesc), ci, C.c_char_p, ci]
        l.tbg_conv2d_f32_variant.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(Epilogue), ci, vp]
        l.tbg_conv2d_bf16_variant.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(Epilogue), ci, vp]
        l.tbg_conv2d_bf16_kernel_name.argtypes = [C.POINTER(ConvDesc), ci, C.c_char_p, ci]
        l.tbg_conv2d_wgrad_bf16_kernel_name.argtypes = [C.POINTER(WgradDesc), C.c_char_p, ci]
        l.tbg_conv2d_bf16.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_wgrad_bf16.argtypes = [C.POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, vp, cf, vp, ll, vp]
        l.tbg_weight_pack_bf16.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        l.tbg_weight_pack_multi.argtypes = [vp, ci, vp]
        l.tbg_modconv_bwd_smalls_f32.argtypes = [vp] * 11 + [ci] * 5 + [vp]
        l.tbg_axpby_planes_f32.argtypes = [vp] * 7 + [ci, ci, vp]
        l.tbg_bias_act_bwd2_f32.argtypes = [vp] * 6 + [ci, ci, ci, C.POINTER(Epilogue), vp]
        l.tbg_torgb_bwd_smalls_f32.argtypes = [vp] * 5 + [ci] * 3 + [cf, ci, vp, vp, vp]
        l.tbg_minibatch_std_fwd_f32.argtypes = [vp, vp, ci, ci, ci, ci, vp]
        l.tbg_minibatch_std_bwd_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
        l.tbg_dense_fwd_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, cf, cf, ci, cf, vp]
        l.tbg_dense_bwd_f32.argtypes = [vp] * 7 + [ci, ci, ci, cf, cf, ci, cf, vp]
        l.tbg_dense_multi_fwd_f32.argtypes = [C.POINTER(DenseItem), ci, ci, ci, cf, cf, cf, vp]
        l.tbg_dense_multi_bwd_f32.argtypes = [C.POINTER(DenseItem), ci, ci, ci, cf, cf, vp]
        l.tbg_weight_pack_bf16_bytes.argtypes = [ci, ci, ci, ci]
        l.tbg_weight_pack_bf16_bytes.restype = C.c_longlong
        l.tbg_conv2d_wgrad_kernel_name.argtypes = [C.POINTER(WgradDesc), C.c_char_p, ci]
        l.tbg_upfirdn2d_kernel_name.argtypes = [ci] * 10 + [C.c_char_p, ci]
        l.tbg_upfirdn2d_f16.argtypes = [vp, vp, vp] + [ci] * 14 + [vp]
        l.tbg_weight_pack_x3_bytes.argtypes = [ci, ci, ci, ci]
        l.tbg_weight_pack_x3_bytes.restype = C.c_longlong
        l.tbg_weight_pack_x3.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        l.tbg_conv2d_x3.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_x3_variant.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(Epilogue), ci, vp]
        l.tbg_conv2d_x3_kernel_name.argtypes = [C.POINTER(ConvDesc), ci, C.c_char_p, ci]
        l.tbg_conv2d_dot_slots.argtypes = [C.POINTER(ConvDesc), ci, ci]
        l.tbg_conv2d_blocks.argtypes = [C.POINTER(ConvDesc), ci, ci]
        l.tbg_conv2d_units.argtypes = [C.POINTER(ConvDesc), vp, ci, vp, vp, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_units_dot_slots.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_small.argtypes = [C.POINTER(ConvDesc), vp, ci, vp, vp, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_units_small_blocks.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_small_dot_slots.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_small_tile_pixels.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_blocks.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_units_s2_bytes.argtypes = [ci] * 5
        l.tbg_units_s2_bytes.restype = C.c_longlong
        l.tbg_units_pack_s2_f32.argtypes = [vp, vp, vp] + [ci] * 7 + [vp]
        l.tbg_upfirdn2d_units_s2_f32.argtypes = [vp, vp, vp, vp] + [ci] * 10 + [vp, ci, vp]
        l.tbg_conv2d_units_s2_blocks.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_s2_tile_channels.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_s2_dot_slots.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_s2.argtypes = [C.POINTER(ConvDesc), vp, ci, vp, vp, C.POINTER(Epilogue), vp]
        l.tbg_conv2d_units_t2_blocks.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_conv2d_units_t2.argtypes = [C.POINTER(ConvDesc), vp, ci, vp, vp, cf, vp]
        l.tbg_conv2d_wgrad_units_s2_workspace_bytes.argtypes = [C.POINTER(WgradDesc)]
        l.tbg_conv2d_wgrad_units_s2_workspace_bytes.restype = C.c_longlong
        l.tbg_conv2d_wgrad_units_s2.argtypes = [C.POINTER(WgradDesc), vp, vp, ci, vp, vp, vp, cf, vp, ll, vp]
        l.tbg_conv2d_units_tile_channels.argtypes = [C.POINTER(ConvDesc), ci]
        l.tbg_bias_act_bwd_units_chunks.argtypes = [ci]
        l.tbg_bias_act_bwd_units.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, C.POINTER(Epilogue), vp]
        l.tbg_units_bytes.argtypes = [ci] * 5
        l.tbg_units_bytes.restype = C.c_longlong
        l.tbg_units_pack_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
        l.tbg_conv2d_wgrad_units_workspace_bytes.argtypes = [C.POINTER(WgradDesc)]
        l.tbg_conv2d_wgrad_units_workspace_bytes.restype = C.c_longlong
        l.tbg_conv2d_wgrad_units.argtypes = [C.POINTER(WgradDesc), vp, vp, ci, vp, vp, vp, cf, vp, ll, vp]
        l.tbg_conv2d_wgrad_x3.argtypes = [C.POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, vp, cf, vp, ll, vp]
        l.tbg_conv2d_wgrad_x3_kernel_name.argtypes = [C.POINTER(WgradDesc), C.c_char_p, ci]
        l.tbg_bias_act_fwd_f32.argtypes = [vp, vp, ci, ci, ci, C.POINTER(Epilogue), vp]
        l.tbg_slab_epilogue_f32.argtypes = [vp, vp, ci, ci, ci, ci, C.POINTER(Epilogue), vp]
        l.tbg_slab_epilogue_units_f32.argtypes = [vp, vp, ci, ci, ci, ci, ci, C.POINTER(Epilogue), vp]
        l.tbg_bias_act_bwd_f32.argtypes = [vp] * 7 + [ci, ci, ci, C.POINTER(Epilogue), vp]
        l.tbg_rgb_project_f32.argtypes = [vp] * 6 + [ci] * 5 + [cf, cf, vp, ci, ci, vp]
        l.tbg_rgb_backproject_f32.argtypes = [vp] * 6 + [ci] * 5 + [cf, vp, ci, ci, vp, vp, vp]
        l.tbg_rgb_backproject_chunks.argtypes = [ci]
        l.tbg_adam_tf_f32.argtypes = [vp, vp, vp, vp, ll, cf, cf, cf, cf, vp, vp]
        l.tbg_ema_lerp_f32.argtypes = [vp, vp, ll, cf, vp]
        l.tbg_demod_coefs_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, cf, vp]
        _lib = _LibProxy(l)
    return _lib


# ----------------------------------------------------------------------------------------
# call recorder (tests): which kernel instantiation every C-ABI call selects
# ----------------------------------------------------------------------------------------
CALL_LOG = None  # None = off; a set collects one key per distinct (entry, instantiation)


class record_calls:
    """``with native.record_calls() as log:`` -- log (a set) receives, for every compute entry called inside the scope, the
    kernel instantiation it selects where the library can name it (convolutions, filter gradients, upfirdn2d: the
    *_kernel_name entries, pure functions of the descriptor) and the entry name otherwise.  Test infrastructure: proves
    that everything a training step launches is also launched by an oracle-compared case."""

    def __enter__(self):
        global CALL_LOG
        self._prev, CALL_LOG = CALL_LOG, set()
        return CALL_LOG

    def __exit__(self, *exc):
        global CALL_LOG
        CALL_LOG = self._prev
        return False


_CONV_FMT = {"tbg_conv2d_f32": 0, "tbg_conv2d_f32_variant": 0, "tbg_conv2d_bf16": 1, "tbg_conv2d_bf16_variant": 1,
             "tbg_conv2d_x3": 2, "tbg_conv2d_x3_variant": 2}
_WGRAD_FMT = {"tbg_conv2d_wgrad_f32": 0, "tbg_conv2d_wgrad_ex_f32": 0, "tbg_conv2d_wgrad_bf16": 1, "tbg_conv2d_wgrad_x3": 2}


def epilogue_opt(epi) -> int:
    """the epilogue instantiation a unit-tensor forward launch with this epilogue runs (third template argument of
    conv_units_fprop_kernel; conv_common.h epi_opt): bit 0 = residual operand, bit 1 = dot / gate operand"""
    return (1 if epi.residual else 0) | (2 if (epi.dot_aux or epi.gate) else 0)


def _call_key(name, a):
    try:
        if name in _CONV_FMT and not name.endswith("_variant"):
            return conv_kernel_name(a[0]._obj, a[4] is not None, _CONV_FMT[name])
        if name in _WGRAD_FMT:
            return wgrad_kernel_name(a[0]._obj, _WGRAD_FMT[name])
        if name == "tbg_conv2d_units":        # d XU planes w y epi stream
            return f"conv_units_fprop_kernel<{a[2]}, {_lib._l.tbg_conv2d_units_tile_channels(a[0], a[2]) // 64}, {epilogue_opt(a[5]._obj)}>"
        if name == "tbg_conv2d_units_s2":
            return f"conv_units_s2_fprop_kernel<{a[2]}, {_lib._l.tbg_conv2d_units_s2_tile_channels(a[0], a[2]) // 64}>"
        if name == "tbg_upfirdn2d_units_s2_f32":
            return f"fir_units_s2_kernel<{a[15]}>"
        if name == "tbg_units_pack_s2_f32":   # x scale U B C Hin Win Ho Wo planes stream
            return f"units_pack_s2_kernel<{a[9]}>"
        if name == "tbg_conv2d_units_t2":
            return f"conv_units_t2_kernel<{a[2]}>"
        if name == "tbg_conv2d_wgrad_units_s2":
            return f"conv_wgrad_units_s2_kernel<{a[3]}>"
        if name == "tbg_conv2d_wgrad_units":  # d SU LU planes ...
            return f"conv_wgrad_units_kernel<{a[3]}>"
        if name == "tbg_bias_act_bwd_units":  # dout out U planes ...
            return f"bias_act_bwd_units_kernel<{a[3]}>"
        if name == "tbg_slab_epilogue_units_f32":  # x y B M H W nslab epi stream
            return f"slab_epilogue_units_kernel<{a[7]._obj.units_planes}>"
        if name == "tbg_units_pack_f32":      # x scale U B C H W planes stream
            # [small]: the input of a small-map launch that no launch of ours produced (<= 2M elements, a 4-6 us launch); the
            # plain steps must not run the producer over anything larger (tests/test_fullwidth_gpu.py)
            return f"units_pack_kernel<{a[7]}>" + (" [small]" if a[3] * a[4] * a[5] * a[6] <= (1 << 21) else "")
        if name == "tbg_conv2d_units_small":  # d XU planes w y epi stream
            d = a[0]._obj
            kw = 3 if (d.KH == 3 and d.KW == 3 and d.sy == 1 and d.sx == 1 and not d.transposed and d.py == 1) else 1
            return f"conv_small_kernel<{a[2]}, {kw}, {_lib._l.tbg_conv2d_units_small_tile_pixels(a[0], a[2]) // 32}>"
        if name in ("tbg_upfirdn2d_f32", "tbg_upfirdn2d_ex_f32", "tbg_upfirdn2d_sep_f32"):
            if name == "tbg_upfirdn2d_f32":      # x k y major inH inW minor kH kW upx upy downx downy padx0 padx1 pady0 pady1
                minor, g = a[6], a[7:17]
            elif name == "tbg_upfirdn2d_ex_f32":  # x k y major inH inW kH kW upx upy downx downy padx0 padx1 pady0 pady1 ...
                minor, g = 1, a[6:16]
            else:                                # x kx ky y major inH inW kH kW upx upy downx downy padx0 padx1 pady0 pady1 ...
                minor, g = 1, a[7:17]
            if name == "tbg_upfirdn2d_sep_f32" and a[19] is not None and a[19]._obj.units_out:  # the blur's unit-sink form
                return f"fir_units_kernel<{a[19]._obj.units_planes}>"
            kH, kW, upx, upy, dnx, dny, px0, _px1, py0, _py1 = g
            buf = C.create_string_buffer(96)
            _lib._l.tbg_upfirdn2d_kernel_name(minor, kH, kW, upx, upy, dnx, dny, px0, py0, int(name.endswith("sep_f32")), buf, 96)
            return buf.value.decode()
    except Exception as e:  # never let the recorder break a call
        return f"{name} [unnamed: {type(e).__name__}]"
    return name


_NOT_COMPUTE = ("tbg_version", "tbg_strerror", "tbg_crc32c", "_kernel_name", "_bytes", "_floats", "_chunks", "_dot_slots", "_blocks", "_tile_channels", "_tile_pixels")  # queries: no device work


class _LibProxy:
    """the ctypes library; while a record_calls scope is active every compute entry also logs what it launches"""

    def __init__(self, l):
        self._l = l

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        if CALL_LOG is None or not name.startswith("tbg_") or name.endswith(_NOT_COMPUTE) or name in _NOT_COMPUTE:
            return fn

        def rec(*a):
            if CALL_LOG is not None:
                CALL_LOG.add(_call_key(name, a))
            return fn(*a)
        return rec


def check(rc: int, what: str):
    if rc != 0:
        raise TbgError(f"{what}: {lib().tbg_strerror(rc).decode()} (code {rc})")


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise TbgError("libtbg_hip kernels need device tensors (no CPU path in the product)")
    if t.dtype not in (torch.float32, torch.int64, torch.bfloat16, torch.float16):
        raise TbgError(f"unsupported dtype {t.dtype}")
    if not t.is_contiguous():
        raise TbgError("tensor must be contiguous")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def epilogue(out_scale=None, bias=None, noise=None, strength=None, residual=None, alpha=1.0, bias_mul=1.0,
             act=ACT_LINEAR, slope=0.2, gain=None, res_scale=1.0, dot_aux=None, dot_out=None, res_first=0, gate=None,
             units_out=None, units_scale=None, units_planes=0) -> Epilogue:
    if gain is None:
        gain = SQRT2 if act == ACT_LRELU else 1.0
    return Epilogue(ptr(out_scale), ptr(bias), ptr(noise), ptr(strength), ptr(residual), ptr(dot_aux), ptr(dot_out), ptr(gate), alpha, bias_mul, slope, gain,
                    res_scale, act, res_first, ptr(units_out), ptr(units_scale), int(units_planes))


def conv_kernel_name(desc: ConvDesc, has_in_scale: bool, fmt=0) -> str:
    """fmt: 0 / False = fp32, 1 / True = bf16, 2 = f32x3 (ops.FMT_*)."""
    buf = C.create_string_buffer(160)
    fn = (lib().tbg_conv2d_kernel_name, lib().tbg_conv2d_bf16_kernel_name, lib().tbg_conv2d_x3_kernel_name)[int(fmt)]
    check(fn(C.byref(desc), int(has_in_scale), buf, 160), "tbg_conv2d_kernel_name")
    return buf.value.decode()


def wgrad_kernel_name(desc: WgradDesc, fmt=0) -> str:
    buf = C.create_string_buffer(160)
    fn = (lib().tbg_conv2d_wgrad_kernel_name, lib().tbg_conv2d_wgrad_bf16_kernel_name,
          getattr(lib(), "tbg_conv2d_wgrad_x3_kernel_name", None))[int(fmt)]
    check(fn(C.byref(desc), buf, 160), "tbg_conv2d_wgrad_kernel_name")
    return buf.value.decode()
