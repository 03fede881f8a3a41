"""ctypes binding of libunetr_hip.so (declared in include/unetr_hip.h).

There is NO fallback: if the shared library is missing or a kernel returns non-zero this raises.  The
library is built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNETR_AMD_LIB: an alternative build of the same ABI (diagnostic builds such as `make x3droplo`); the product loads libunetr_hip.so
LIB_PATH = os.environ.get("UNETR_AMD_LIB") or os.path.join(_HERE, "libunetr_hip.so")

PREC_F32 = 0
PREC_BF16 = 1
PREC_BF16X3 = 2      # fp32 storage, operands split into bf16 (hi, lo) pairs inside the kernels (csrc/common.hpp: PrecBF16x3)
ABI_VERSION = 17       # = UNETR_ABI_VERSION of include/unetr_hip.h this table of signatures was written against

_ERR = {1: "invalid argument", 2: "kernel launch failed", 3: "unsupported shape/configuration",
        4: "workspace too small"}


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int),
        ("a_trans", c_int), ("b_trans", c_int),
        ("lda", c_long), ("ldb", c_long), ("ldc", c_long),
        ("strideA", c_long), ("strideB", c_long), ("strideC", c_long),
        ("bias", c_void_p), ("res", c_void_p),
        ("ldr", c_long), ("strideR", c_long), ("res_mod", c_int),
        ("pre", c_void_p), ("aux", c_void_p), ("ldaux", c_long),
        ("act", c_int), ("accumulate", c_int), ("alpha", c_float), ("prec", c_int), ("b_x3words", c_int),
    ]


class GemmBf16Desc(ctypes.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int), ("b_kn", c_int),
        ("lda", c_long), ("ldb", c_long), ("ldc", c_long), ("ldcb", c_long),
        ("bias", c_void_p), ("res", c_void_p), ("ldr", c_long), ("res_mod", c_int),
        ("pre", c_void_p), ("aux", c_void_p), ("ldaux", c_long),
        ("act", c_int), ("accumulate", c_int), ("alpha", c_float),
        ("tc_d", c_int), ("tc_h", c_int), ("tc_w", c_int), ("tc_cout", c_int), ("x3", c_int),
    ]


class LnGemmDesc(ctypes.Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_long), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
        ("W", c_void_p), ("ldw", c_long), ("bias", c_void_p), ("act", c_int),
        ("pre", c_void_p), ("ldpre", c_long), ("Cb", c_void_p), ("ldcb", c_long), ("C", c_void_p), ("ldc", c_long),
        ("xn", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int),
    ]


class GroupedProblem(ctypes.Structure):
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("dw", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int)]


class AdamWArena(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("m", c_void_p), ("v", c_void_p), ("shadow_bf16", c_void_p), ("steps", c_void_p),
                ("total", c_long), ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
                ("shadow_x3", c_void_p)]


class PackProblem(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("Cin", c_int), ("Cout", c_int), ("kind", c_int)]


class ReduceProblem(ctypes.Structure):
    _fields_ = [("part", c_void_p), ("dst", c_void_p), ("n", c_long), ("rows", c_int)]


class SplitProblem(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_long), ("cols", c_long), ("second", c_int)]


class ColsumProblem(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("ld", c_long), ("M", c_int), ("N", c_int), ("x_bf16", c_int)]


P = c_void_p
_SIGNATURES = {
    "unetr_abi_version": [],
    "unetr_gemm": [ctypes.POINTER(GemmDesc), P, P, P, P, c_size_t, P],
    "unetr_gemm_bf16": [ctypes.POINTER(GemmBf16Desc), P, P, P, P, P, c_size_t, P],
    "unetr_gemm_bf16_ln_bwd": [ctypes.POINTER(GemmBf16Desc), P, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P, c_size_t, P],
    "unetr_gemm_bf16_ln_fwd": [ctypes.POINTER(GemmBf16Desc), P, P, P, P, P, c_float, P, P, P, P, P, c_size_t, P],
    "unetr_cast_bf16": [P, P, c_long, P],
    "unetr_split_words": [P, P, c_long, P],
    "unetr_split_stack_bf16": [P, P, c_long, c_long, c_int, P],
    "unetr_split_stack_bf16_grouped": [ctypes.POINTER(SplitProblem), c_int, P],
    "unetr_ln_gemm_bf16": [ctypes.POINTER(LnGemmDesc), P],
    "unetr_attention_bf16_fwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "unetr_attention_bf16_bwd": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "unetr_gemm_grouped_wgrad": [ctypes.POINTER(GroupedProblem), c_int, c_int, P],
    "unetr_gemm_bf16_grouped_wgrad": [ctypes.POINTER(GroupedProblem), c_int, P],
    "unetr_gemm_bf16_grouped_wgrad_adamw": [ctypes.POINTER(GroupedProblem), c_int, ctypes.POINTER(AdamWArena), ctypes.POINTER(c_int), P],
    "unetr_adamw_ranges": [ctypes.POINTER(AdamWArena), P, c_int, c_long, P],
    "unetr_gemm_bf16_grouped_wgrad_bf16out": [ctypes.POINTER(GroupedProblem), c_int, P, P, c_long, P],
    "unetr_cast_bf16_ranges": [P, P, P, c_int, c_long, P],
    "unetr_colsum_grouped": [ctypes.POINTER(ColsumProblem), c_int, P],
    "unetr_tconv_fwd": [P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_tconv_dgrad": [P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_tconv_wgrad": [P, c_long, P, c_long, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_tconv2_fwd": [P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_tconv2_wgrad": [P, c_long, P, c_long, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_tconv2_dgrad": [P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_tconv2_fwd_supported": [c_long, c_int, c_int, c_long, c_long],
    "unetr_tconv2_wgrad_supported": [c_long, c_int, c_int, c_long, c_long],
    "unetr_pixel_shuffle2": [P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_pixel_unshuffle2_bf16": [P, c_long, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_colsum": [P, c_long, c_int, c_int, P, c_int, P, c_size_t, P],
    "unetr_layernorm_fwd": [P, P, P, P, P, P, P, c_int, c_int, c_float, P],
    "unetr_layernorm_bwd": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, c_size_t, P],
    "unetr_attention_fwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P],
    "unetr_attention_bwd": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P],
    "unetr_conv_pack_weight": [P, P, c_int, c_int, c_int, c_int, P],
    "unetr_conv_gemm_fwd": [P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_conv_gemm_wgrad": [P, c_long, P, c_long, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_conv3_pack_weight": [P, P, c_int, c_int, c_int, c_int, P],
    "unetr_conv3_fwd": [P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_conv3_fwd_fused": [P, c_long, P, P, c_long, P, P, P, c_long, P, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_conv3_fwd_parts": [P, c_long, P, P, c_long, P, P, P, c_long, P, ctypes.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_conv3_dgrad_stats": [P, c_long, P, P, c_long, P, c_long, P, P, ctypes.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_instnorm_apply_fin": [P, c_long, P, c_int, P, c_long, P, c_int, P, P, c_float, P, c_long, c_int, c_long, c_int, c_int, c_int, P],
    "unetr_instnorm_apply_fin_img": [P, c_long, P, c_int, P, c_int, P, P, c_int, P, P, c_float, P, c_long, c_int, c_long, c_int, c_int, c_int, P],
    "unetr_instnorm_bwd_img": [P, c_long, P, c_long, P, P, c_int, P, P, P, c_long, P, ctypes.POINTER(c_int), c_int, c_long, c_int, c_int, P, c_size_t, c_int, P],
    "unetr_instnorm_bwd_apply_fin": [P, c_long, P, c_long, P, P, c_long, P, P, c_int, c_int, P, c_long, P, c_long, c_int, c_long, c_int, c_int, c_int, P],
    "unetr_conv3_wgrad_parts": [P, c_long, P, c_long, P, c_long, P, c_size_t, ctypes.POINTER(c_long), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_tconv2_wgrad_parts": [P, c_long, P, c_long, P, c_size_t, ctypes.POINTER(c_long), c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_reduce_rows_grouped": [ctypes.POINTER(ReduceProblem), c_int, P],
    "unetr_conv3_dgrad_fused": [P, c_long, P, P, c_long, P, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_conv3_pack_grouped": [ctypes.POINTER(PackProblem), c_int, c_int, P],
    "unetr_conv3_pack_1x1": [P, P, c_int, c_int, c_int, P],
    "unetr_instnorm_stats_finalize": [P, c_int, c_int, c_long, c_int, c_float, P, P],
    "unetr_conv3_wgrad": [P, c_long, P, c_long, P, P, c_long, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P],
    "unetr_debug_tr16": [P, P, P],
    "unetr_instnorm_stats": [P, c_long, c_int, c_long, c_int, c_float, P, P, c_size_t, c_int, P],
    "unetr_instnorm_apply": [P, c_long, P, P, c_long, P, P, c_long, c_int, c_long, c_int, c_int, c_int, P],
    "unetr_instnorm_bwd": [P, c_long, P, c_long, P, P, c_long, P, P, c_long, P, c_long, c_int, c_long, c_int, c_int, P, c_size_t, c_int, P],
    "unetr_nchw_to_nhwc": [P, P, c_long, c_int, c_int, c_long, c_int, P],
    "unetr_nhwc_to_nchw": [P, c_long, P, c_int, c_int, c_long, c_int, c_int, P],
    "unetr_patch_gather": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_counter_add": [P, P, c_int, P],
    "unetr_add_cast_bf16": [P, P, P, P, c_long, P],
    "unetr_copy_rows": [P, c_long, P, c_long, c_long, c_int, c_int, c_int, P],
    "unetr_outconv_fwd": [P, c_long, P, P, P, c_int, c_long, c_int, c_int, c_int, P],
    "unetr_outconv_bwd": [P, P, c_long, P, P, c_long, P, P, c_int, c_long, c_int, c_int, P, c_size_t, c_int, P],
    "unetr_outconv_in_fwd": [P, c_long, P, c_int, P, c_long, P, c_int, P, P, c_float, P, P, P, c_int, c_long, c_int, c_int, c_int, P],
    "unetr_outconv_in_bwd": [P, P, c_long, P, P, c_long, P, P, P, c_long, P, P, P, c_int, c_long, c_int, c_int, P, c_size_t, c_int, P],
    "unetr_dicece_fwd": [P, P, c_int, c_int, c_long, c_int, c_float, c_float, P, P, P, c_size_t, P],
    "unetr_dicece_bwd": [P, P, P, P, P, c_int, c_int, c_long, c_int, P],
    "unetr_sw_accumulate": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "unetr_sw_finalize": [P, P, c_int, c_int, c_long, P],
    "unetr_dice_counts": [P, P, c_int, c_int, c_long, c_int, P, P, c_size_t, P],
    "unetr_ranking_loss_fwd": [P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P, P, P, c_size_t, P],
    "unetr_ranking_loss_bwd": [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "unetr_adamw": [P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_float, P, P, P],
    "unetr_adamw_reduced": [P, P, c_int, c_float, P, P, c_long, c_float, c_float, c_float, c_float, c_float, P, P, P, P],
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES) + ("unetr_conv3_packed_bytes", "unetr_conv3_packed_1x1_bytes", "unetr_ranking_workspace_floats",
                                            "unetr_conv3_wgrad_rows", "unetr_tconv2_wgrad_rows", "unetr_outconv_in_bwd_rows")

_lib = None


def load():
    """Load libunetr_hip.so once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C {os.path.join(_HERE, 'csrc')}`. There is no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.unetr_abi_version.restype = c_int
    got = lib.unetr_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports C-ABI version {got}, this binding was written against {ABI_VERSION}: the "
                           f"library is stale -- rebuild the extension (`make -C {os.path.join(_HERE, 'csrc')}`).")
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.unetr_conv3_packed_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.unetr_conv3_packed_bytes.restype = c_size_t
    lib.unetr_conv3_packed_1x1_bytes.argtypes = [c_int, c_int, c_int]
    lib.unetr_conv3_packed_1x1_bytes.restype = c_size_t
    lib.unetr_conv3_wgrad_rows.argtypes = [c_int] * 9
    lib.unetr_conv3_wgrad_rows.restype = c_long
    lib.unetr_tconv2_wgrad_rows.argtypes = [c_int] * 6
    lib.unetr_tconv2_wgrad_rows.restype = c_long
    lib.unetr_outconv_in_bwd_rows.argtypes = [c_int, c_long, c_int, c_int]
    lib.unetr_outconv_in_bwd_rows.restype = c_long
    lib.unetr_ranking_workspace_floats.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.unetr_ranking_workspace_floats.restype = c_size_t
    _lib = lib
    return lib


def call_rc(name, *args):
    """like call(), but hands 'unsupported shape' (3) back to the caller, which then takes the unfused route"""
    rc = getattr(load(), name)(*args)
    if rc not in (0, 3):
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
    return rc


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
