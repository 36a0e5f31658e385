"""ctypes binding of liblssvc_hip.so (the C ABI declared in include/lssvc_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises -- loudly -- instead of silently computing elsewhere.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: PyTorch-ROCm carries its own libamdhip64; whichever HIP runtime is loaded first serves
#                the whole process, and device memory / streams handed to this library come from torch's. Loading
#                liblssvc_hip.so before torch binds the system runtime instead ("no ROCm-capable device is detected").

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSSVC_HIP_LIB") or os.path.join(_HERE, "lib", "liblssvc_hip.so")     # override: A/B of two builds

CONV_MAX_INPUTS = 3
CONV_CK = 8

ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2
INACT_NONE, INACT_LRELU, INACT_SQUARE = 0, 1, 2
PREC_F32, PREC_F16X3 = 0, 1
PREC_MASK, PREC_SPLIT_IN = 0xff, 0x100
EPI_NONE, EPI_X_MUL_RSQRT, EPI_X_MUL_SQRT, EPI_X_DIV_SQRT = 0, 1, 2, 3


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("ld", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("inp", View * CONV_MAX_INPUTS), ("n_in", C.c_int32),
        ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("Cout", C.c_int32), ("M_pad", C.c_int32),
        ("in_act", C.c_int32), ("in_slope", C.c_float),
        ("epilogue", C.c_int32), ("gdn_x", View),
        ("act", C.c_int32), ("slope", C.c_float),
        ("residual", View), ("out_scale", C.c_float),
        ("pixel_shuffle", C.c_int32), ("out", View),
        ("precision", C.c_int32), ("weight16", C.c_void_p), ("weight16_unscale", C.c_float),
        ("residual2", View),
    ]


class FfnDesc(C.Structure):
    _fields_ = [
        ("x", View), ("pre_in", View), ("pre_w16", C.c_void_p), ("pre_unscale", C.c_float), ("pre_bias", C.c_void_p),
        ("ident", View), ("w1_16", C.c_void_p), ("w1_unscale", C.c_float), ("b1", C.c_void_p), ("hidden", C.c_int32),
        ("w2_16", C.c_void_p), ("w2_unscale", C.c_float), ("b2", C.c_void_p), ("slope", C.c_float), ("out", View), ("skip", View),
    ]


class CdfTable(C.Structure):
    _fields_ = [("cdfs", C.c_void_p), ("n_cdfs", C.c_int32), ("stride", C.c_int32), ("sizes", C.c_void_p),
                ("offsets", C.c_void_p)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class PrepSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("name", C.c_char * 120), ("name2", C.c_char * 120), ("splits", C.c_int32 * 3), ("n_splits", C.c_int32),
                ("flag", C.c_int32)]


PREP_CONV, PREP_CONV_F16X3, PREP_CONVT, PREP_DWCONV, PREP_GDN, PREP_VECTOR, PREP_BIT_ESTIMATOR, PREP_ENTROPY_BOTTLENECK, PREP_FFN_F16X3 = range(1, 10)
PREP_MAX_BLOBS = 8

VP = C.POINTER(View)
TP = C.POINTER(CdfTable)

# name -> (restype, argtypes); every symbol include/lssvc_hip.h declares
SIGNATURES = {
    "lssvc_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "lssvc_conv2d_variant": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "lssvc_conv2d_last_kernel": (C.c_char_p, []),
    "lssvc_conv1x1_dw3x3_f16x3": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_ffn_f16x3": (C.c_int, [C.POINTER(FfnDesc), C.c_void_p]),
    "lssvc_ffn_f16x3_lds_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "lssvc_ffn_f16x3_is_streamed": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "lssvc_dwconv3x3": (C.c_int, [VP, C.c_void_p, C.c_void_p, VP, C.c_void_p]),
    "lssvc_resize_bilinear": (C.c_int, [VP, VP, C.c_float, C.c_void_p]),
    "lssvc_flow_warp": (C.c_int, [VP, VP, VP, C.c_void_p]),
    "lssvc_pool2x2": (C.c_int, [VP, VP, C.c_int32, C.c_void_p]),
    "lssvc_softmax2_blend": (C.c_int, [VP, VP, VP, VP, C.c_void_p]),
    "lssvc_spynet_prep": (C.c_int, [VP, VP, VP, VP, C.c_void_p]),
    "lssvc_avgpool_pyramid3": (C.c_int, [VP, VP, VP, VP, C.c_void_p]),
    "lssvc_add": (C.c_int, [VP, VP, VP, C.c_void_p]),
    "lssvc_copy": (C.c_int, [VP, VP, C.c_void_p]),
    "lssvc_lrelu": (C.c_int, [VP, VP, C.c_float, C.c_void_p]),
    "lssvc_presplit": (C.c_int, [VP, VP, C.c_int32, C.c_float, C.c_void_p]),
    "lssvc_pad_crop": (C.c_int, [VP, VP, C.c_int32, C.c_int32, C.c_void_p]),
    "lssvc_absmax": (C.c_int, [VP, C.c_void_p, C.c_void_p]),
    "lssvc_fill_zero": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "lssvc_clamp_inplace": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "lssvc_offset_diversity": (C.c_int, [VP, VP, VP, C.c_void_p, C.c_void_p, VP, C.c_void_p]),
    "lssvc_nchw_to_nhwc": (C.c_int, [C.c_void_p, VP, C.c_void_p]),
    "lssvc_nhwc_to_nchw": (C.c_int, [VP, C.c_void_p, C.c_void_p]),
    "lssvc_reduce_workspace_bytes": (C.c_int64, []),
    "lssvc_laplace_quant_bits": (C.c_int, [VP, VP, VP, VP, VP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_four_part_step": (C.c_int, [VP, VP, VP, C.POINTER(C.c_int32), VP, VP, VP, C.c_void_p]),
    "lssvc_laplace_bits": (C.c_int, [VP, VP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_factorized_quant_bits": (C.c_int, [VP, C.c_void_p, VP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_gaussian_conditional": (C.c_int, [VP, VP, VP, VP, VP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_entropy_bottleneck": (C.c_int, [VP, C.c_void_p, VP, VP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_export_symbols": (C.c_int, [VP, VP, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_import_symbols": (C.c_int, [C.c_void_p, VP, C.c_void_p, C.c_void_p, VP, C.c_void_p]),
    "lssvc_export_symbols_i16": (C.c_int, [VP, VP, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_import_symbols_i16": (C.c_int, [C.c_void_p, VP, C.c_void_p, C.c_void_p, VP, C.c_void_p]),
    "lssvc_build_indexes": (C.c_int, [VP, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "lssvc_yuv420_to_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, VP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_rgb8_to_frame": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, VP, C.c_void_p]),
    "lssvc_resample2d": (C.c_int, [VP, VP, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    "lssvc_rgb_to_yuv420": (C.c_int, [VP, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_sqdiff_sum": (C.c_int, [VP, VP, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_sqdiff_sum_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_rans_encoder_new": (C.c_void_p, []),
    "lssvc_rans_encoder_free": (None, [C.c_void_p]),
    "lssvc_rans_encoder_reset": (None, [C.c_void_p]),
    "lssvc_rans_encode_with_indexes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, TP]),
    "lssvc_rans_encode_with_indexes_i16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, TP]),
    "lssvc_rans_encoder_flush": (C.c_int64, [C.c_void_p]),
    "lssvc_rans_encoder_bytes": (C.c_void_p, [C.c_void_p]),
    "lssvc_rans_decoder_new": (C.c_void_p, []),
    "lssvc_rans_decoder_free": (None, [C.c_void_p]),
    "lssvc_rans_decoder_set_stream": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "lssvc_rans_decode_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, TP, C.c_void_p]),
    "lssvc_rans_decode_stream_i16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, TP, C.c_void_p]),
    "lssvc_pmf_to_quantized_cdf": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lssvc_prepare_weights": (C.c_int, [C.POINTER(Tensor), C.c_int32, C.POINTER(PrepSpec), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "lssvc_engine_create": (C.c_void_p, [C.c_int32]),
    "lssvc_engine_destroy": (None, [C.c_void_p]),
    "lssvc_engine_load_checkpoint": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Tensor), C.c_int32]),
    "lssvc_engine_load_intra": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lssvc_engine_load_inter": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "lssvc_engine_set_scale": (C.c_int, [C.c_void_p, C.c_float, C.c_int32, C.c_int32]),
    "lssvc_engine_iframe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lssvc_engine_pframe": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.POINTER(C.c_double)] + [C.c_void_p] * 7),
    "lssvc_engine_load_inter_layers": (C.c_int, [C.c_void_p] + [C.c_char_p] * 4),
    "lssvc_engine_pframe_lookahead": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7 + [C.POINTER(C.c_double)] + [C.c_void_p] * 7),
    "lssvc_engine_lookahead_reset": (C.c_int, [C.c_void_p]),
    "lssvc_engine_plan_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "lssvc_engine_plan_meta": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_int64)]),
    "lssvc_engine_load_stream": (C.c_int, [C.c_void_p] + [C.c_char_p] * 6),
    "lssvc_engine_encode_iframe": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)] + [C.c_void_p] * 4),
    "lssvc_engine_decode_iframe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64] + [C.c_void_p] * 4),
    "lssvc_engine_encode_pframe": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)] + [C.c_void_p] * 5),
    "lssvc_engine_decode_pframe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64] + [C.c_void_p] * 9),
    "lssvc_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "lssvc_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "lssvc_last_error": (C.c_char_p, []),
    "lssvc_version": (C.c_int, []),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "liblssvc_hip.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C lssvc_amd/csrc). There is no CPU fallback for the LSSVC hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class LssvcHipError(RuntimeError):
    pass


def check(status):
    if status != 0:
        raise LssvcHipError(lib.lssvc_last_error().decode("utf-8", "replace"))
