"""ctypes binding of libbridgelang_hip.so (the C ABI declared in include/bridgelang_hip.h).

The library is the product: there is NO CPU or PyTorch fallback. If the shared object is missing or an entry point is
absent, importing/using the ops raises immediately (the driver records which .so files GPU tests actually loaded).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libbridgelang_hip.so"

BL_OK, BL_E_SHAPE, BL_E_ALIGN, BL_E_LAUNCH, BL_E_ARG = 0, -1, -2, -3, -4
_ERR = {BL_E_SHAPE: "BL_E_SHAPE (unsupported shape)", BL_E_ALIGN: "BL_E_ALIGN (alignment)",
        BL_E_LAUNCH: "BL_E_LAUNCH (HIP launch failed)", BL_E_ARG: "BL_E_ARG (null pointer / bad enum)"}

# enum bl_epilogue
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_RES, EPI_SWIGLU, EPI_F32, EPI_F32_BF16R = range(8)
EPI_SWIGLU_KEEP, EPI_BIAS_GELU_KEEP, EPI_SWIGLU_BWD, EPI_GELU_BWD = range(8, 12)     # training-step forms (bl_gemm_bf16)


class GemmDesc(C.Structure):
    """struct bl_gemm_desc — field order and types mirror include/bridgelang_hip.h exactly."""
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("epilogue", C.c_int32),
        ("bias", C.c_void_p),
        ("scale", C.c_void_p),
        ("res", C.c_void_p), ("ldres", C.c_int64),
        ("res_row_mod", C.c_int32),
        ("out_group", C.c_int32), ("out_stride", C.c_int32), ("out_offset", C.c_int32),
        ("a_norm_weight", C.c_void_p),
        ("a_norm_eps", C.c_float),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("C2", C.c_void_p),
        ("ldc2", C.c_int64),
    ]


class BatchOpDesc(C.Structure):
    """One entry of bl_batched_ops' device table (struct BatchOp in csrc/train.hip)."""
    _fields_ = [("kind", C.c_int32), ("nblocks", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("rows_pad", C.c_int32),
                ("bx_count", C.c_int32), ("ld", C.c_int64), ("kt_total", C.c_int64), ("kb_off", C.c_int64), ("n", C.c_int64),
                ("src", C.c_void_p), ("dst", C.c_void_p), ("scale", C.c_float), ("pad", C.c_int32)]


class AttnDesc(C.Structure):
    """struct bl_attn_desc."""
    _fields_ = [
        ("q", C.c_void_p), ("q_bs", C.c_int64), ("q_hs", C.c_int64), ("q_rs", C.c_int64),
        ("k", C.c_void_p), ("k_bs", C.c_int64), ("k_hs", C.c_int64), ("k_rs", C.c_int64),
        ("v", C.c_void_p), ("v_bs", C.c_int64), ("v_hs", C.c_int64), ("v_rs", C.c_int64),
        ("o", C.c_void_p), ("o_bs", C.c_int64), ("o_hs", C.c_int64), ("o_rs", C.c_int64),
        ("key_mask", C.c_void_p), ("mask_bs", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Sq", C.c_int32), ("Skv", C.c_int32), ("head_dim", C.c_int32),
        ("causal", C.c_int32),
        ("scale", C.c_float),
    ]


_vp, _i32, _i64, _u32, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float

# name -> (restype, argtypes): every symbol include/bridgelang_hip.h declares
SIGNATURES = {
    "bl_abi_version": (C.c_int, []),
    "bl_build_arch": (C.c_char_p, []),
    "bl_fill_synth_bf16": (C.c_int, [_vp, _i64, _u32, _f32, _f32, _vp]),
    "bl_fill_synth_bf16_2d": (C.c_int, [_vp, _i64, _i64, _i64, _u32, _f32, _f32, _vp]),
    "bl_pack_weight_bf16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bl_gemm_fp8": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp]),
    "bl_quantize_rows_fp8": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "bl_colamax_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "bl_transpose_quantize_fp8": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    "bl_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "bl_gemm_skinny_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "bl_gemm_skinny_rows_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "bl_gemm_tn_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "bl_rmsnorm_skinny_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "bl_layernorm_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "bl_rmsnorm_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "bl_attention_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "bl_attention_rope_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "bl_attention_lse_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp, _vp]),
    "bl_attention_backward_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bl_attention_decode_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "bl_attention_decode_rope_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp, _vp, _i32, _vp]),
    "bl_attention_decode_rope_pos_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp, _vp, _i32, _vp, _vp]),
    "bl_attention_decode_rope_grouped_bf16": (C.c_int, [C.POINTER(AttnDesc), _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "bl_rope_kvcache_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "bl_embed_splice_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "bl_argmax_f32": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "bl_cross_entropy_f32": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "bl_cross_entropy_backward_f32": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp]),
    "bl_rmsnorm_backward_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "bl_layernorm_backward_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "bl_colsum_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "bl_swiglu_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "bl_swiglu_backward_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "bl_gelu_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "bl_gelu_backward_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "bl_rope_backward_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "bl_transpose_pad_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "bl_pack_weight_into_bf16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp]),
    "bl_transpose_pack_into_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i32, _i64, _i64, _vp]),
    "bl_transpose_pack_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "bl_sumsq_partial_f32": (C.c_int, [_vp, _i64, _vp, _i32, _vp]),
    "bl_clip_coef_f32": (C.c_int, [_vp, _i32, _f32, _vp, _vp]),
    "bl_adamw_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "bl_rope_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "bl_scale_residual_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "bl_layerscale_backward_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp]),
    "bl_gemm_tn_small_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _i32, _f32, _vp, _i64, _vp]),
    "bl_scale_bf16": (C.c_int, [_vp, _f32, _vp, _i64, _vp]),
    "bl_lora_block_mask_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "bl_axpy_f32": (C.c_int, [_vp, _vp, _f32, _i64, _vp]),
    "bl_cast_f32_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "bl_cast_bf16_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "bl_memset_zero": (C.c_int, [_vp, _i64, _vp]),
    "bl_copy_bytes": (C.c_int, [_vp, _vp, _i64, _vp]),
    "bl_batched_ops": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "bl_dropout_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _f32, _vp, _u32, _vp, _i64, _vp]),
    "bl_dropout_grad_fix_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _f32, _vp, _u32, _vp, _i64, _vp]),
    "bl_map_rows_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "bl_embed_backward_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "bl_preprocess_u8_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "bl_resample_pass_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "bl_crop_resize_bilinear_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    "bl_im2col_patch14_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _vp]),
    "bl_write_prefix_tokens_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp]),
}

_lib = None


class BridgeLangHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once). Raises if it has not been built — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("BRIDGELANG_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise BridgeLangHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C bridgelang_amd/csrc`). bridgelang_amd has no CPU/PyTorch fallback."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing → loud failure
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != BL_OK:
        raise BridgeLangHipError(f"{what} failed: {_ERR.get(rc, rc)}")
