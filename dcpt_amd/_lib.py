"""ctypes binding of libdcpt_hip.so (C ABI in include/dcpt_hip.h).

The library is the product: there is NO CPU / eager fallback.  If it cannot be loaded, or a tensor
is not on a HIP device, the callers raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdcpt_hip.so")   # the in-tree build is the only library the product loads

ABI_VERSION = 15
_lib = None
_lock = threading.Lock()

f32p = C.c_void_p  # device pointers travel as integers
i64 = C.c_int64
cint = C.c_int
sz = C.c_size_t
stream_t = C.c_void_p

_PARAM_FIELDS = [
    "norm1_w", "norm1_b", "conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "conv3_b",
    "sca_w", "sca_b", "norm2_w", "norm2_b", "conv4_w", "conv4_b", "conv5_w", "conv5_b", "beta", "gamma",
]
_SAVED_FIELDS = ["t1", "t2", "y", "v", "mu1", "rstd1", "mu2", "rstd2", "pooled", "s", "xn1", "xn2", "g"]


class NafBlockParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PARAM_FIELDS]


class NafBlockGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PARAM_FIELDS]


class NafBlockSaved(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _SAVED_FIELDS]


MDTA_PARAM_FIELDS = ("norm_w", "norm_b", "qkv_w", "dw_w", "proj_w", "temperature")
MDTA_SAVED_FIELDS = ("mu", "rstd", "qkv1", "qkv", "nrm", "ghat", "attn", "attnT", "out_att", "xn")
GDFN_PARAM_FIELDS = ("norm_w", "norm_b", "in_w", "dw_w", "out_w")
GDFN_SAVED_FIELDS = ("mu", "rstd", "u", "t", "xn")


class NafBlockSavedBf16(C.Structure):   # same field order as NafBlockSaved; the [M][.] tensors are bf16
    _fields_ = [(n, C.c_void_p) for n in _SAVED_FIELDS]


class MdtaParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in MDTA_PARAM_FIELDS]


class MdtaSaved(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in MDTA_SAVED_FIELDS]


class GdfnParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GDFN_PARAM_FIELDS]


class GdfnSaved(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GDFN_SAVED_FIELDS]


class BneckGroup(C.Structure):   # dcpt_bneck_group_t: one conv -> LayerNorm group of the classifier head's BottleneckBlock
    _fields_ = [("w", C.c_void_p), ("wpacked", C.c_void_p), ("wpacked_bytes", C.c_size_t), ("lnw", C.c_void_p), ("lnb", C.c_void_p),
                ("z", C.c_void_p), ("y", C.c_void_p), ("mu", C.c_void_p), ("rstd", C.c_void_p), ("dw", C.c_void_p), ("dlnw", C.c_void_p),
                ("dlnb", C.c_void_p)]


PARAM_FIELDS = tuple(_PARAM_FIELDS)
SAVED_FIELDS = tuple(_SAVED_FIELDS)

# name -> (restype, argtypes); mirrors include/dcpt_hip.h one to one
SIGNATURES = {
    "dcpt_last_error": (C.c_char_p, []),
    "dcpt_abi_version": (cint, []),
    "dcpt_ln2d_fwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, i64, cint, C.c_float, stream_t]),
    "dcpt_ln2d_bwd_ws_bytes": (sz, [i64, cint]),
    "dcpt_ln2d_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, i64, cint, stream_t]),
    "dcpt_nafblock_fwd_ws_bytes": (sz, [cint, cint, cint, cint]),
    "dcpt_nafblock_bwd_ws_bytes": (sz, [cint, cint, cint, cint]),
    "dcpt_nafblock_fwd": (cint, [C.POINTER(NafBlockParams), f32p, f32p, C.POINTER(NafBlockSaved), C.c_void_p, sz,
                                 cint, cint, cint, cint, stream_t]),
    "dcpt_nafblock_bwd": (cint, [C.POINTER(NafBlockParams), C.POINTER(NafBlockGrads), f32p, C.POINTER(NafBlockSaved),
                                 f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_nafblock_fwd_bf16_ws_bytes": (sz, [cint, cint, cint, cint]),
    "dcpt_nafblock_bwd_bf16_ws_bytes": (sz, [cint, cint, cint, cint]),
    "dcpt_nafblock_fwd_bf16": (cint, [C.POINTER(NafBlockParams), f32p, f32p, C.POINTER(NafBlockSavedBf16), C.c_void_p, sz,
                                      cint, cint, cint, cint, stream_t]),
    "dcpt_nafblock_bwd_bf16": (cint, [C.POINTER(NafBlockParams), C.POINTER(NafBlockGrads), f32p, C.POINTER(NafBlockSavedBf16),
                                      f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_set_gemm_x3": (cint, [C.c_void_p, sz, cint]),
    "dcpt_gemm_x3_scratch_misses": (C.c_longlong, []),
    "dcpt_nafblock_wpack_bf16_bytes": (sz, [cint]),
    "dcpt_nafblock_bf16_fused_ffn": (cint, [cint]),
    "dcpt_nafblock_fused_ffn": (cint, [cint]),
    "dcpt_nafblock_wpack_bf16": (cint, [C.POINTER(NafBlockParams), C.c_void_p, sz, cint, stream_t]),
    "dcpt_nafblock_wpack_bf16_multi": (cint, [C.POINTER(NafBlockParams), C.POINTER(C.c_void_p), C.POINTER(sz), C.POINTER(cint), cint, stream_t]),
    "dcpt_nafblock_fwd_bf16_packed": (cint, [C.POINTER(NafBlockParams), C.c_void_p, sz, f32p, f32p, C.POINTER(NafBlockSavedBf16), C.c_void_p, sz,
                                             cint, cint, cint, cint, stream_t]),
    "dcpt_nafblock_bwd_bf16_packed": (cint, [C.POINTER(NafBlockParams), C.c_void_p, sz, C.POINTER(NafBlockGrads), f32p,
                                             C.POINTER(NafBlockSavedBf16), f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_conv1x1_wgrad_bf16_ws_bytes": (sz, [i64, cint, cint]),
    "dcpt_conv1x1_wgrad_bf16": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, i64, cint, cint, stream_t]),
    "dcpt_cast_f32_bf16": (cint, [f32p, f32p, i64, stream_t]),
    "dcpt_cast_bf16_f32": (cint, [f32p, f32p, i64, stream_t]),
    "dcpt_conv_ln_bf16_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint, cint]),
    "dcpt_conv_ln_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, cint, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint,
                                     cint, cint, cint, stream_t]),
    "dcpt_conv_ln_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz,
                                     cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv_ln_bwd_acc_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz,
                                         cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv1x1_pool_relu_bf16_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint]),
    "dcpt_conv1x1_pool_relu_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv1x1_pool_relu_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint,
                                               stream_t]),
    "dcpt_conv_wpack_bf16_bytes": (sz, [cint, cint, cint]),
    "dcpt_conv_wpack_bf16_multi": (cint, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(sz), C.POINTER(cint), C.POINTER(cint),
                                          C.POINTER(cint), cint, stream_t]),
    "dcpt_conv_ln_fwd_bf16_packed": (cint, [f32p, f32p, C.c_void_p, sz, f32p, f32p, f32p, cint, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint,
                                            cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv_ln_bwd_acc_bf16_packed": (cint, [f32p, f32p, f32p, C.c_void_p, sz, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p,
                                                f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv1x1_pool_relu_fwd_bf16_packed": (cint, [f32p, f32p, C.c_void_p, sz, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint,
                                                      stream_t]),
    "dcpt_conv1x1_pool_relu_bwd_bf16_packed": (cint, [f32p, f32p, f32p, C.c_void_p, sz, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint,
                                                      cint, cint, stream_t]),
    "dcpt_conv3x3_in_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv3x3_in_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv3x3_out_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv3x3_out_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_down2x2_bf16_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_down2x2_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_down2x2_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_down2x2_bwd_acc_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_up_ps_bf16_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_up_ps_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_up_ps_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_nafblock_local_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint]),
    "dcpt_nafblock_local_fwd": (cint, [C.POINTER(NafBlockParams), f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, cint,
                                       stream_t]),
    "dcpt_conv3x3_in_fwd": (cint, [f32p, f32p, f32p, f32p, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv3x3_in_bwd_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_conv3x3_in_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv3x3_out_fwd": (cint, [f32p, f32p, f32p, f32p, f32p, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv3x3_out_bwd_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_conv3x3_out_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_down2x2_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_down2x2_fwd": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_down2x2_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_down2x2_bwd_acc": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_up_ps_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_up_ps_fwd": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_up_ps_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_fused_bias_act": (cint, [f32p, f32p, f32p, f32p, i64, cint, i64, cint, cint, C.c_float, C.c_float, stream_t]),
    "dcpt_conv_ln_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint, cint]),
    "dcpt_conv_ln_fwd": (cint, [f32p, f32p, f32p, f32p, f32p, cint, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint,
                                cint, cint, cint, stream_t]),
    "dcpt_conv_ln_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz,
                                cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv_ln_bwd_acc": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz,
                                    cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv1x1_pool_relu_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint]),
    "dcpt_conv1x1_pool_relu_fwd": (cint, [f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv1x1_pool_relu_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint,
                                          stream_t]),
    "dcpt_mix_fwd": (cint, [f32p, f32p, f32p, cint, cint, f32p, i64, stream_t]),
    "dcpt_mix_bwd_ws_bytes": (sz, [i64]),
    "dcpt_mix_bwd": (cint, [f32p, f32p, f32p, cint, cint, f32p, f32p, C.c_void_p, sz, i64, stream_t]),
    "dcpt_meanpool_fc_ws_bytes": (sz, [cint, cint, cint]),
    "dcpt_meanpool_fc_fwd": (cint, [f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_meanpool_fc_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_mix_fwd_bf16": (cint, [f32p, f32p, f32p, cint, cint, f32p, i64, stream_t]),
    "dcpt_mix_bwd_bf16": (cint, [f32p, f32p, f32p, cint, cint, f32p, f32p, C.c_void_p, sz, i64, stream_t]),
    "dcpt_meanpool_fc_fwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_meanpool_fc_bwd_bf16": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_patch_unfold": (cint, [f32p, f32p, cint, cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_patch_fold": (cint, [f32p, f32p, cint, cint, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint, cint]),
    "dcpt_conv_fwd": (cint, [f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_conv_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_mdta_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint]),
    "dcpt_mdta_fwd": (cint, [C.POINTER(MdtaParams), f32p, f32p, C.POINTER(MdtaSaved), C.c_void_p, sz, cint, cint, cint, cint,
                             cint, cint, stream_t]),
    "dcpt_mdta_bwd": (cint, [C.POINTER(MdtaParams), C.POINTER(MdtaParams), f32p, C.POINTER(MdtaSaved), f32p, f32p, C.c_void_p,
                             sz, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_gdfn_ws_bytes": (sz, [cint, cint, cint, cint, cint, cint]),
    "dcpt_gdfn_fwd": (cint, [C.POINTER(GdfnParams), f32p, f32p, C.POINTER(GdfnSaved), C.c_void_p, sz, cint, cint, cint, cint,
                             cint, cint, stream_t]),
    "dcpt_gdfn_bwd": (cint, [C.POINTER(GdfnParams), C.POINTER(GdfnParams), f32p, C.POINTER(GdfnSaved), f32p, f32p, C.c_void_p,
                             sz, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_prompt_mix_fwd": (cint, [f32p, f32p, f32p, f32p, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_prompt_mix_bwd_ws_bytes": (sz, [cint, cint, cint]),
    "dcpt_prompt_mix_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, C.c_void_p, sz, cint, cint, cint, cint, cint, cint, stream_t]),
    "dcpt_pixel_unshuffle": (cint, [f32p, f32p, cint, cint, cint, cint, stream_t]),
    "dcpt_pixel_shuffle": (cint, [f32p, f32p, cint, cint, cint, cint, stream_t]),
    "dcpt_concat_channels": (cint, [f32p, f32p, f32p, i64, cint, cint, stream_t]),
    "dcpt_split_channels": (cint, [f32p, f32p, f32p, i64, cint, cint, stream_t]),
    "dcpt_bottleneck_bf16_ws_bytes": (sz, [cint, cint, cint, cint, cint]),
    "dcpt_bottleneck_fwd_bf16": (cint, [f32p, C.POINTER(BneckGroup), C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_bottleneck_bwd_bf16": (cint, [f32p, f32p, C.POINTER(BneckGroup), f32p, C.c_void_p, sz, cint, cint, cint, cint, stream_t]),
    "dcpt_trace_enable": (cint, [cint]),
    "dcpt_trace_read": (sz, [C.c_char_p, sz]),
    "dcpt_prof_enable": (cint, [cint]),
    "dcpt_prof_read": (cint, [C.POINTER(C.c_double), cint]),
    "dcpt_set_side_stream": (cint, [cint]),
    "dcpt_adamw_step": (cint, [cint, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int64), C.c_void_p, stream_t]),
    "dcpt_allreduce_flat": (cint, [f32p, sz, C.c_void_p, C.c_float, stream_t]),
    "dcpt_nchw_to_nhwc": (cint, [f32p, f32p, cint, cint, cint, stream_t]),
    "dcpt_nhwc_to_nchw": (cint, [f32p, f32p, cint, cint, cint, stream_t]),
}


class DcptHipError(RuntimeError):
    pass


def lib_exists() -> bool:
    return os.path.exists(LIB_PATH)


def load():
    """Load (once) and return the ctypes handle.  Raises DcptHipError if the .so is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            # the library is built in-tree; if this checkout has no binary yet, compile it now (hipcc, gfx950).
            # There is still no CPU / eager fallback: without hipcc this raises.
            try:
                from . import build as _build

                _build.build()
            except Exception as e:  # noqa: BLE001
                raise DcptHipError(
                    f"{LIB_PATH} not found and could not be built ({e}); run `python -m dcpt_amd.build` (hipcc, gfx950). "
                    "dcpt_amd has no CPU/eager fallback."
                ) from e
        # One HIP runtime per process: make sure torch's bundled libamdhip64 is the one already
        # loaded before our library's DT_NEEDED libamdhip64.so.7 is resolved (loading ours first would
        # bring in /opt/rocm's copy and torch's streams/pointers would belong to another runtime).
        import torch

        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.dcpt_abi_version() != ABI_VERSION:
            raise DcptHipError("libdcpt_hip.so ABI version mismatch")
        _lib = lib
        return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dcpt_last_error()
        raise DcptHipError(f"{what}: rc={rc}: {msg.decode() if msg else ''}")
