"""Thin torch-tensor front end over the C ABI (bridgelang_amd/_lib.py → libbridgelang_hip.so).

PyTorch is plumbing only: tensors give device memory and the current HIP stream; every arithmetic operation is a
hand-written gfx950 kernel behind `bl_*`. Each builder returns an `Op` (a prepared C call whose descriptor structs are
built once); `Op.run()` enqueues it on the current stream. The engine keeps lists of Ops and replays them (directly or
under HIP-graph capture), so steady-state host cost is one ctypes call per kernel.
"""
from __future__ import annotations

import functools
import math

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (AttnDesc, GemmDesc, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_KEEP, EPI_BIAS_RES, EPI_F32, EPI_F32_BF16R, EPI_GELU_BWD,
                   EPI_NONE, EPI_SWIGLU_BWD, EPI_SWIGLU_KEEP,
                   EPI_RES, EPI_SWIGLU)

__all__ = ["Op", "gemm", "gemm_fp8", "quantize_rows_fp8", "quantize_weight_fp8", "pack_weight", "unpack_weight", "cross_entropy", "layernorm", "rmsnorm", "rmsnorm_skinny", "skinny_rows_supported", "attention", "attention_rope", "attention_decode", "attention_decode_rope", "attention_decode_rope_grouped", "skinny_supported", "rope_kvcache", "embed_splice",
           "argmax", "im2col_patch14", "preprocess_u8", "resample_coeffs", "resize_bicubic_u8", "resize_u8", "crop_resize_bilinear_u8", "write_prefix_tokens", "fill_synth", "run_all",
           "EPI_NONE", "EPI_BIAS", "EPI_BIAS_GELU", "EPI_BIAS_RES", "EPI_RES", "EPI_SWIGLU", "EPI_F32", "EPI_F32_BF16R",
           "EPI_SWIGLU_KEEP", "EPI_BIAS_GELU_KEEP", "EPI_SWIGLU_BWD", "EPI_GELU_BWD"]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _bf16(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.bfloat16 or not t.is_cuda:
        raise TypeError(f"{what}: expected a CUDA/HIP bfloat16 tensor, got {t.dtype} on {t.device}")
    return t


def _rows(t: torch.Tensor, what: str) -> int:
    """Leading dimension (elements) of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{what}: expected a 2-D tensor with contiguous last dim, got shape {tuple(t.shape)} "
                         f"stride {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


class Op:
    """A prepared call into libbridgelang_hip.so. Keeps its tensors alive."""
    __slots__ = ("name", "fn", "args", "keep", "flops", "bytes")

    def __init__(self, name: str, fn, args: tuple, keep: tuple, flops: float = 0.0, nbytes: float = 0.0):
        self.name, self.fn, self.args, self.keep = name, fn, args, keep
        self.flops, self.bytes = flops, nbytes   # ALGORITHMIC work of this launch (bench.py roofline accounting)

    def run(self, stream: Optional[int] = None) -> None:
        rc = self.fn(*self.args, stream if stream is not None else _stream())
        if rc != 0:
            _lib.check(rc, self.name)


def glue(name: str, fn, keep: tuple = ()) -> Op:
    """A host-side index/copy step (torch data movement on the current stream, no arithmetic) wrapped as an Op so it can
    sit inside an op plan; `fn()` runs when the plan reaches it."""
    def call(stream):
        fn()
        return 0
    return Op(name, call, (), tuple(keep))


def run_all(ops: Sequence[Op]) -> None:
    s = _stream()
    for op in ops:
        rc = op.fn(*op.args, s)
        if rc != 0:
            _lib.check(rc, op.name)


# ---------------------------------------------------------------------------------------------------------------
def pack_weight(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear-layout weight [N, K] (last dim contiguous) → fragment-major [N/16, K/32, 64, 8] (bl_pack_weight_bf16).
    N % 16 == 0 and K % 32 == 0 (zero-pad first). Done once at load time."""
    lib = _lib.load()
    _bf16(w, "w")
    N, K = w.shape
    if out is None:
        out = torch.empty(N // 16, K // 32, 64, 8, dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.bl_pack_weight_bf16(w.data_ptr(), _rows(w, "w"), N, K, _bf16(out, "out").data_ptr(), _stream()),
               "bl_pack_weight_bf16")
    return out


def unpack_weight(wp: torch.Tensor) -> torch.Tensor:
    """Inverse of pack_weight (torch glue; used only when exporting a state dict)."""
    nt, ks = wp.shape[0], wp.shape[1]
    return wp.view(nt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(nt * 16, ks * 32)


SKINNY_K = (512, 1024, 1536, 4096, 5120, 11008, 13824)
_SKINNY_EPI = (EPI_NONE, EPI_RES, EPI_SWIGLU, EPI_F32, EPI_F32_BF16R)


def skinny_supported(M: int, K: int, epilogue: int) -> bool:
    """Shapes the weight-streaming kernel is instantiated for (gemm_skinny.hip::launch_skinny)."""
    return M <= 16 and K in SKINNY_K and epilogue in _SKINNY_EPI


def skinny_rows_supported(M: int, K: int, epilogue: int) -> bool:
    """Shapes bl_gemm_skinny_rows_bf16 takes: up to 128 stacked rows in the skinny kernel's summation order."""
    return M <= 128 and K in SKINNY_K and epilogue in _SKINNY_EPI


def gemm(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE, *,
         bias: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, res_row_mod: int = 0,
         out_map: Optional[Tuple[int, int, int]] = None,
         skinny: Optional[bool] = None, algo_nk: Optional[Tuple[int, int]] = None,
         a_norm: Optional[Tuple[torch.Tensor, float]] = None, workspace: Optional[torch.Tensor] = None,
         skinny_rows: bool = False, out2: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    """out = epilogue(A @ W.T).  A [M,K] row-major activations; W = PACKED weight [N/16, K/32, 64, 8] (pack_weight);
    out [rows, N] (N/2 for SWIGLU, 2N for SWIGLU_BWD; fp32 for F32*).

    Training forms: EPI_SWIGLU_KEEP / EPI_BIAS_GELU_KEEP write the pre-activation to `out` and the activation to `out2`
    ([rows, N/2] / [rows, N]); EPI_SWIGLU_BWD / EPI_GELU_BWD apply the activation's backward to the product, reading the
    saved pre-activation through `res` ([rows, 2N] gate/up pairs / [rows, N]).

    `out_map=(group, stride, offset)` remaps output rows (see bl_gemm_desc).
    `skinny=None` picks the weight-streaming kernel automatically for M <= 16 when it supports K.
    `algo_nk=(N, K)` gives the un-padded logical sizes for FLOP accounting when N or K carry zero padding.
    `a_norm=(weight, eps)` fuses HF LlamaRMSNorm on the rows of A (skinny kernel only: M <= 16).
    `workspace`: optional scratch tensor (>= 64 MiB) enabling the split-K tail of the 256x256 kernel.
    `skinny_rows=True`: bl_gemm_skinny_rows_bf16 — up to 128 rows in the skinny kernel's arithmetic (row results
    bit-identical to the M <= 16 kernel; the merged decode iteration of StaggeredDecodePipeline).
    """
    lib = _lib.load()
    _bf16(A, "A"); _bf16(W, "W")
    if W.dim() != 4 or W.shape[2:] != (64, 8) or not W.is_contiguous():
        raise ValueError(f"gemm: W must be a packed weight [N/16, K/32, 64, 8] (ops.pack_weight), got {tuple(W.shape)}")
    M, K = A.shape
    N, Kw = W.shape[0] * 16, W.shape[1] * 32
    if Kw != K:
        raise ValueError(f"gemm: K mismatch A{tuple(A.shape)} W(packed) K={Kw}")
    want = torch.float32 if epilogue in (EPI_F32, EPI_F32_BF16R) else torch.bfloat16
    if out.dtype != want or not out.is_cuda:
        raise TypeError(f"gemm: out must be {want} on device")
    d = GemmDesc()
    d.A, d.lda = A.data_ptr(), _rows(A, "A")
    d.W, d.ldw = W.data_ptr(), K
    d.C, d.ldc = out.data_ptr(), _rows(out, "out")
    d.M, d.N, d.K, d.epilogue = M, N, K, epilogue
    keep = [A, W, out]
    if bias is not None:
        d.bias = _bf16(bias, "bias").data_ptr(); keep.append(bias)
    if scale is not None:
        d.scale = _bf16(scale, "scale").data_ptr(); keep.append(scale)
    if res is not None:
        d.res, d.ldres = _bf16(res, "res").data_ptr(), _rows(res, "res"); keep.append(res)
    d.res_row_mod = res_row_mod
    if epilogue in (EPI_SWIGLU_KEEP, EPI_BIAS_GELU_KEEP):
        if out2 is None:
            raise ValueError("gemm: the *_KEEP epilogues need out2")
        n2 = N // 2 if epilogue == EPI_SWIGLU_KEEP else N
        if out2.shape[1] < n2 or out2.shape[0] < M:
            raise ValueError(f"gemm: out2 {tuple(out2.shape)} too small for [{M}, {n2}]")
        d.C2, d.ldc2 = _bf16(out2, "out2").data_ptr(), _rows(out2, "out2"); keep.append(out2)
    elif out2 is not None:
        raise ValueError("gemm: out2 only with the *_KEEP epilogues")
    if epilogue in (EPI_SWIGLU_BWD, EPI_GELU_BWD) and (res is None or res.shape[0] < M or
                                                       res.shape[1] < (2 * N if epilogue == EPI_SWIGLU_BWD else N)):
        raise ValueError("gemm: the *_BWD epilogues read the saved pre-activation through res")
    if out_map is not None:
        d.out_group, d.out_stride, d.out_offset = out_map
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
        keep.append(workspace)
    n_out = N // 2 if epilogue == EPI_SWIGLU else 2 * N if epilogue == EPI_SWIGLU_BWD else N
    if out.shape[1] < n_out:
        raise ValueError(f"gemm: out has {out.shape[1]} columns, needs {n_out}")
    if out_map is None and out.shape[0] < M:
        raise ValueError(f"gemm: out has {out.shape[0]} rows, needs {M}")
    use_skinny = skinny
    if use_skinny is None:
        use_skinny = skinny_supported(M, K, epilogue)
    if a_norm is not None:
        if not use_skinny:
            raise ValueError("gemm: a_norm (fused RMSNorm) needs the skinny kernel (M <= 16 and a supported K)")
        d.a_norm_weight, d.a_norm_eps = _bf16(a_norm[0], "a_norm weight").data_ptr(), float(a_norm[1])
        keep.append(a_norm[0])
    if skinny_rows:
        if not skinny_rows_supported(M, K, epilogue) or a_norm is not None or out_map is not None:
            raise ValueError(f"gemm: skinny_rows needs M <= 128, K in {SKINNY_K}, no a_norm / out_map (M={M}, K={K})")
        name = "bl_gemm_skinny_rows_bf16"
    else:
        name = "bl_gemm_skinny_bf16" if use_skinny else "bl_gemm_bf16"
    fn = getattr(lib, name)
    # algorithmic work: logical FLOPs; bytes = each operand once + the output once
    esz = 4 if epilogue in (EPI_F32, EPI_F32_BF16R) else 2
    op = Op(name, fn, (C.byref(d),), (d, *keep),
            flops=2.0 * M * (algo_nk[0] if algo_nk else N) * (algo_nk[1] if algo_nk else K),
            nbytes=2.0 * (M * K + N * K) + esz * M * n_out)
    if run:
        op.run()
    return op


# ---- FP8 (e4m3) GEMM family — BASELINE configs[4]; no reference counterpart -----------------------------------------
FP8_MAX = 448.0


def quantize_weight_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """nn.Linear weight [N, K] (bf16, N % 16 == 0, K % 128 == 0) → (packed e4m3 weight, fp32 scale [N]): per output
    channel scale = amax / 448, RNE cast, then the fragment-major packing of pack_weight on the bytes viewed as bf16
    pairs (bl_gemm_fp8 reads the same byte layout as bl_gemm_bf16). Load-time work, torch ops."""
    N, K = w.shape
    if N % 16 or K % 128:
        raise ValueError("quantize_weight_fp8: N % 16 == 0 and K % 128 == 0")
    wf = w.float()
    amax = wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    packed = pack_weight(q.view(torch.uint8).view(N, K // 2, 2).view(torch.bfloat16).reshape(N, K // 2).contiguous())
    return packed, scale.contiguous()


def quantize_rows_fp8(x: torch.Tensor, q: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None,
                      run: bool = True):
    """Activations bf16 [rows, cols] → (e4m3 codes uint8 [rows, cols], fp32 scale per row = amax / 448)
    (bl_quantize_rows_fp8). Returns (q, scales, op)."""
    lib = _lib.load()
    _bf16(x, "x")
    rows, cols = x.shape
    if q is None:
        q = torch.empty(rows, cols, dtype=torch.uint8, device=x.device)
    if scales is None:
        scales = torch.empty(rows, dtype=torch.float32, device=x.device)
    op = Op("bl_quantize_rows_fp8", lib.bl_quantize_rows_fp8,
            (x.data_ptr(), _rows(x, "x"), rows, cols, q.data_ptr(), q.stride(0), scales.data_ptr()), (x, q, scales),
            nbytes=3.0 * rows * cols)
    if run:
        op.run()
    return q, scales, op


def gemm_fp8(A8: torch.Tensor, scale_a: torch.Tensor, W8: torch.Tensor, scale_w: torch.Tensor, out: torch.Tensor,
             epilogue: int = EPI_NONE, *, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
             run: bool = True) -> Op:
    """out = epilogue(scale_a[m] · scale_w[n] · (A8 @ W8.T)) on the fp8 MFMA path (bl_gemm_fp8). A8 uint8 e4m3 codes
    [M, K]; W8 = quantize_weight_fp8(...)[0]."""
    lib = _lib.load()
    if A8.dtype != torch.uint8 or not A8.is_cuda or A8.stride(1) != 1:
        raise TypeError("gemm_fp8: A8 must be a CUDA/HIP uint8 matrix of e4m3 codes")
    if W8.dim() != 4 or W8.shape[2:] != (64, 8) or not W8.is_contiguous():
        raise ValueError("gemm_fp8: W8 must come from quantize_weight_fp8")
    M, K = A8.shape
    N, Kw = W8.shape[0] * 16, W8.shape[1] * 64
    if Kw != K:
        raise ValueError(f"gemm_fp8: K mismatch A{tuple(A8.shape)} W K={Kw}")
    want = torch.float32 if epilogue in (EPI_F32, EPI_F32_BF16R) else torch.bfloat16
    if out.dtype != want or scale_a.dtype != torch.float32 or scale_w.dtype != torch.float32:
        raise TypeError("gemm_fp8: out dtype / fp32 scales")
    if scale_a.numel() < M or scale_w.numel() != N:
        raise ValueError("gemm_fp8: scale_a [M], scale_w [N]")
    d = GemmDesc()
    d.A, d.lda = A8.data_ptr(), A8.stride(0)
    d.W, d.ldw = W8.data_ptr(), K
    d.C, d.ldc = out.data_ptr(), _rows(out, "out")
    d.M, d.N, d.K, d.epilogue = M, N, K, epilogue
    keep = [A8, W8, out, scale_a, scale_w]
    if bias is not None:
        d.bias = _bf16(bias, "bias").data_ptr(); keep.append(bias)
    if res is not None:
        d.res, d.ldres = _bf16(res, "res").data_ptr(), _rows(res, "res"); keep.append(res)
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    esz = 4 if epilogue in (EPI_F32, EPI_F32_BF16R) else 2
    op = Op("bl_gemm_fp8", lib.bl_gemm_fp8, (C.byref(d), scale_a.data_ptr(), scale_w.data_ptr()), (d, *keep),
            flops=2.0 * M * N * K, nbytes=1.0 * (M * K + N * K) + esz * M * n_out)
    if run:
        op.run()
    return op


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, eps: float = 1e-6,
              run: bool = True) -> Op:
    lib = _lib.load()
    _bf16(x, "x"); _bf16(out, "out")
    rows, dim = x.shape
    op = Op("bl_layernorm_bf16", lib.bl_layernorm_bf16,
            (x.data_ptr(), _rows(x, "x"), _bf16(w, "w").data_ptr(), _bf16(b, "b").data_ptr(), out.data_ptr(),
             _rows(out, "out"), rows, dim, float(eps)), (x, w, b, out))
    if run:
        op.run()
    return op


def rmsnorm(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, eps: float = 1e-6, run: bool = True) -> Op:
    lib = _lib.load()
    _bf16(x, "x"); _bf16(out, "out")
    rows, dim = x.shape
    op = Op("bl_rmsnorm_bf16", lib.bl_rmsnorm_bf16,
            (x.data_ptr(), _rows(x, "x"), _bf16(w, "w").data_ptr(), out.data_ptr(), _rows(out, "out"), rows, dim,
             float(eps)), (x, w, out))
    if run:
        op.run()
    return op


def rmsnorm_skinny(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, eps: float = 1e-6, run: bool = True) -> Op:
    """HF LlamaRMSNorm in the arithmetic of the skinny GEMM's fused a_norm (bl_rmsnorm_skinny_bf16); dim in SKINNY_K."""
    lib = _lib.load()
    _bf16(x, "x"); _bf16(out, "out")
    rows, dim = x.shape
    op = Op("bl_rmsnorm_skinny_bf16", lib.bl_rmsnorm_skinny_bf16,
            (x.data_ptr(), _rows(x, "x"), _bf16(w, "w").data_ptr(), out.data_ptr(), _rows(out, "out"), rows, dim,
             float(eps)), (x, w, out), nbytes=4.0 * rows * dim)
    if run:
        op.run()
    return op


def _attn_desc(q, k, v, o, B, H, Sq, Skv, hd, q_str, k_str, v_str, o_str, causal, scale, key_mask):
    d = AttnDesc()
    d.q, (d.q_bs, d.q_hs, d.q_rs) = _bf16(q, "q").data_ptr(), q_str
    d.k, (d.k_bs, d.k_hs, d.k_rs) = _bf16(k, "k").data_ptr(), k_str
    d.v, (d.v_bs, d.v_hs, d.v_rs) = _bf16(v, "v").data_ptr(), v_str
    d.o, (d.o_bs, d.o_hs, d.o_rs) = _bf16(o, "o").data_ptr(), o_str
    if key_mask is not None:
        if key_mask.dtype != torch.uint8 or key_mask.dim() != 2 or key_mask.stride(1) != 1:
            raise TypeError("key_mask must be a [B, Skv] uint8 tensor")
        d.key_mask, d.mask_bs = key_mask.data_ptr(), key_mask.stride(0)
    d.B, d.H, d.Sq, d.Skv, d.head_dim, d.causal, d.scale = B, H, Sq, Skv, hd, int(causal), float(scale)
    return d


def attention(q, k, v, o, *, B: int, H: int, Sq: int, Skv: int, head_dim: int, q_strides, k_strides, v_strides,
              o_strides, causal: bool, scale: Optional[float] = None, key_mask: Optional[torch.Tensor] = None,
              run: bool = True) -> Op:
    """Flash attention over strided [batch, head, row, head_dim] views; strides are (batch, head, row) in elements."""
    lib = _lib.load()
    scale = head_dim ** -0.5 if scale is None else scale
    d = _attn_desc(q, k, v, o, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal, scale,
                   key_mask)
    op = Op("bl_attention_bf16", lib.bl_attention_bf16, (C.byref(d),), (d, q, k, v, o, key_mask))
    if run:
        op.run()
    return op


def attention_rope(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, o: torch.Tensor, cos: torch.Tensor,
                   sin: torch.Tensor, *, B: int, S: int, H: int, head_dim: int, pos0: int = 0, scale: Optional[float] = None,
                   key_mask: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    """Prefill attention over fused, un-rotated qkv rows [B*S, 3*H*hd] with RoPE and the KV-cache write fused in
    (bl_attention_rope_bf16): caches [B, H, cache_len, hd] receive rotated k and v at positions pos0.., o [B*S, H*hd]."""
    lib = _lib.load()
    D = H * head_dim
    for t, n in ((qkv, "qkv"), (k_cache, "k_cache"), (v_cache, "v_cache"), (cos, "cos"), (sin, "sin")):
        _bf16(t, n)
        if not t.is_contiguous():
            raise ValueError(f"attention_rope: {n} must be contiguous")
    cache_len = k_cache.shape[2]
    scale = head_dim ** -0.5 if scale is None else scale
    st = (S * 3 * D, head_dim, 3 * D)
    d = _attn_desc(qkv, qkv[:, D:], qkv[:, 2 * D:], o, B, H, S, S, head_dim, st, st, st, (S * D, head_dim, D), True, scale, key_mask)
    op = Op("bl_attention_rope_bf16", lib.bl_attention_rope_bf16,
            (C.byref(d), cos.data_ptr(), sin.data_ptr(), pos0, k_cache.data_ptr(), v_cache.data_ptr(), cache_len),
            (d, qkv, k_cache, v_cache, o, cos, sin, key_mask))
    if run:
        op.run()
    return op


def attention_decode(q, k, v, o, *, B: int, H: int, Skv: int, head_dim: int, q_strides, k_strides, v_strides,
                     o_strides, scale: Optional[float] = None, key_mask: Optional[torch.Tensor] = None,
                     run: bool = True) -> Op:
    lib = _lib.load()
    scale = head_dim ** -0.5 if scale is None else scale
    d = _attn_desc(q, k, v, o, B, H, 1, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, False, scale,
                   key_mask)
    op = Op("bl_attention_decode_bf16", lib.bl_attention_decode_bf16, (C.byref(d),), (d, q, k, v, o, key_mask))
    if run:
        op.run()
    return op


def attention_decode_rope(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, o: torch.Tensor,
                          cos: torch.Tensor, sin: torch.Tensor, *, B: int, H: int, head_dim: int, pos: int,
                          scale: Optional[float] = None, key_mask: Optional[torch.Tensor] = None,
                          rope_pos: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    """One decode step's attention with RoPE and the KV-cache append fused: qkv [B, 3*H*hd] is the step's fused
    projection (q | k | v), caches [B, H, cache_len, hd]; rotates q and k at `pos`, writes k', v to cache row `pos`,
    attends over keys 0..pos, writes o [B, H*hd]."""
    lib = _lib.load()
    D = H * head_dim
    for t, n in ((qkv, "qkv"), (k_cache, "k_cache"), (v_cache, "v_cache"), (cos, "cos"), (sin, "sin")):
        _bf16(t, n)
        if not t.is_contiguous():
            raise ValueError(f"attention_decode_rope: {n} must be contiguous")
    cache_len = k_cache.shape[2]
    if pos >= cache_len or pos >= cos.shape[0]:
        raise ValueError("attention_decode_rope: position outside the cache / rope table")
    scale = head_dim ** -0.5 if scale is None else scale
    cs = (H * cache_len * head_dim, cache_len * head_dim, head_dim)
    d = _attn_desc(qkv, k_cache, v_cache, o, B, H, 1, pos + 1, head_dim, (3 * D, head_dim, 3 * D), cs, cs,
                   (D, head_dim, D), False, scale, key_mask)
    if rope_pos is not None:     # right-padded prompts: per-sequence rotation position (int32 [B]), shared cache row
        if rope_pos.dtype != torch.int32 or not rope_pos.is_cuda or rope_pos.numel() != B or not rope_pos.is_contiguous():
            raise TypeError("attention_decode_rope: rope_pos must be a contiguous CUDA/HIP int32 tensor [B]")
        op = Op("bl_attention_decode_rope_pos_bf16", lib.bl_attention_decode_rope_pos_bf16,
                (C.byref(d), cos.data_ptr(), sin.data_ptr(), pos, rope_pos.data_ptr()),
                (d, qkv, k_cache, v_cache, o, cos, sin, key_mask, rope_pos))
        if run:
            op.run()
        return op
    op = Op("bl_attention_decode_rope_bf16", lib.bl_attention_decode_rope_bf16,
            (C.byref(d), cos.data_ptr(), sin.data_ptr(), pos), (d, qkv, k_cache, v_cache, o, cos, sin, key_mask))
    if run:
        op.run()
    return op


def rope_kvcache(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, k_cache: torch.Tensor,
                 v_cache: torch.Tensor, *, B: int, S: int, H: int, head_dim: int, pos0: int, run: bool = True) -> Op:
    """qkv [B*S, 3*H*hd] (q rotated in place); caches [B, H, cache_len, hd]; cos/sin [max_pos, hd/2] bf16."""
    lib = _lib.load()
    for t, n in ((qkv, "qkv"), (cos, "cos"), (sin, "sin"), (k_cache, "k_cache"), (v_cache, "v_cache")):
        _bf16(t, n)
        if not t.is_contiguous():
            raise ValueError(f"rope_kvcache: {n} must be contiguous")
    cache_len = k_cache.shape[2]
    if pos0 + S > cos.shape[0]:
        raise ValueError("rope_kvcache: position exceeds the cos/sin table")
    op = Op("bl_rope_kvcache_bf16", lib.bl_rope_kvcache_bf16,
            (qkv.data_ptr(), B, S, H, head_dim, cos.data_ptr(), sin.data_ptr(), pos0, k_cache.data_ptr(),
             v_cache.data_ptr(), cache_len), (qkv, cos, sin, k_cache, v_cache))
    if run:
        op.run()
    return op


def embed_splice(ids: torch.Tensor, table: torch.Tensor, dst: torch.Tensor, n_patches: int, run: bool = True) -> Op:
    """dst[b,0] = table[ids[b,0]]; dst[b, 1+n_patches+j] = table[ids[b,1+j]].  ids int64 [B,L]; dst [B, L+n_patches, D]."""
    lib = _lib.load()
    if ids.dtype != torch.int64 or not ids.is_contiguous():
        raise TypeError("ids must be a contiguous int64 tensor")
    B, L = ids.shape
    dim = table.shape[1]
    if tuple(dst.shape) != (B, L + n_patches, dim) or not dst.is_contiguous():
        raise ValueError(f"embed_splice: dst must be contiguous [{B}, {L + n_patches}, {dim}]")
    op = Op("bl_embed_splice_bf16", lib.bl_embed_splice_bf16,
            (ids.data_ptr(), B, L, _bf16(table, "table").data_ptr(), dim, n_patches, _bf16(dst, "dst").data_ptr()),
            (ids, table, dst))
    if run:
        op.run()
    return op


def argmax(logits: torch.Tensor, out: torch.Tensor, run: bool = True) -> Op:
    lib = _lib.load()
    if logits.dtype != torch.float32 or out.dtype != torch.int64:
        raise TypeError("argmax: logits fp32 [rows, n], out int64 [rows]")
    rows, n = logits.shape
    op = Op("bl_argmax_f32", lib.bl_argmax_f32, (logits.data_ptr(), _rows(logits, "logits"), rows, n, out.data_ptr()),
            (logits, out))
    if run:
        op.run()
    return op


def cross_entropy(logits: torch.Tensor, targets: torch.Tensor, row_loss: torch.Tensor, mean_and_count: torch.Tensor,
                  ignore_index: int = -100, run: bool = True) -> Op:
    """Per-row CE over fp32 logits [rows, n] against int64 targets [rows] (already shifted); mean over valid rows."""
    lib = _lib.load()
    if logits.dtype != torch.float32 or targets.dtype != torch.int64 or row_loss.dtype != torch.float32:
        raise TypeError("cross_entropy: logits/row_loss fp32, targets int64")
    rows, n = logits.shape
    op = Op("bl_cross_entropy_f32", lib.bl_cross_entropy_f32,
            (logits.data_ptr(), _rows(logits, "logits"), rows, n, targets.data_ptr(), ignore_index,
             row_loss.data_ptr(), mean_and_count.data_ptr()), (logits, targets, row_loss, mean_and_count))
    if run:
        op.run()
    return op


def im2col_patch14(pixel_values: torch.Tensor, chan0: int, out: torch.Tensor, run: bool = True) -> Op:
    """pixel_values [B,6,224,224] bf16 → out [B*256, ld>=588] patch rows of channels chan0..chan0+2."""
    lib = _lib.load()
    _bf16(pixel_values, "pixel_values"); _bf16(out, "out")
    if tuple(pixel_values.shape[1:]) != (6, 224, 224) or not pixel_values.is_contiguous():
        raise ValueError("im2col_patch14: pixel_values must be contiguous [B, 6, 224, 224]")
    B = pixel_values.shape[0]
    op = Op("bl_im2col_patch14_bf16", lib.bl_im2col_patch14_bf16,
            (pixel_values.data_ptr(), B, chan0, out.data_ptr(), _rows(out, "out")), (pixel_values, out))
    if run:
        op.run()
    return op


def write_prefix_tokens(prefix: torch.Tensor, x: torch.Tensor, B: int, T: int, run: bool = True) -> Op:
    lib = _lib.load()
    n_prefix, dim = prefix.shape
    op = Op("bl_write_prefix_tokens_bf16", lib.bl_write_prefix_tokens_bf16,
            (_bf16(prefix, "prefix").data_ptr(), n_prefix, dim, _bf16(x, "x").data_ptr(), B, T), (prefix, x))
    if run:
        op.run()
    return op


def attention_decode_rope_grouped(qkv: torch.Tensor, k_caches: Sequence[torch.Tensor], v_caches: Sequence[torch.Tensor],
                                  o: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, *, B: int, H: int, head_dim: int,
                                  pos: Sequence[int], run: bool = True) -> Op:
    """len(pos) decode iterations of different batches in one launch: rows g*B.. of qkv [G*B, 3*H*hd] / o [G*B, H*hd] use
    caches k_caches[g] / v_caches[g] and position pos[g] (bl_attention_decode_rope_grouped_bf16)."""
    lib = _lib.load()
    G, D = len(pos), H * head_dim
    if not (1 <= G <= 8) or len(k_caches) != G or len(v_caches) != G:
        raise ValueError("attention_decode_rope_grouped: 1..8 groups, one cache pair and position each")
    for t, n in [(qkv, "qkv"), (cos, "cos"), (sin, "sin")] + [(t, "cache") for t in list(k_caches) + list(v_caches)]:
        _bf16(t, n)
        if not t.is_contiguous():
            raise ValueError(f"attention_decode_rope_grouped: {n} must be contiguous")
    cache_len = k_caches[0].shape[2]
    if any(tuple(t.shape) != tuple(k_caches[0].shape) for t in list(k_caches) + list(v_caches)):
        raise ValueError("attention_decode_rope_grouped: all caches must share one shape")
    if max(pos) >= cache_len or max(pos) >= cos.shape[0] or qkv.shape[0] != G * B or o.shape[0] != G * B:
        raise ValueError("attention_decode_rope_grouped: position outside the cache / rope table, or row count != G*B")
    cs = (H * cache_len * head_dim, cache_len * head_dim, head_dim)
    d = _attn_desc(qkv, k_caches[0], v_caches[0], o, B, H, 1, pos[0] + 1, head_dim, (3 * D, head_dim, 3 * D), cs, cs,
                   (D, head_dim, D), False, head_dim ** -0.5, None)
    kp = (C.c_void_p * G)(*[t.data_ptr() for t in k_caches])
    vp = (C.c_void_p * G)(*[t.data_ptr() for t in v_caches])
    pp = (C.c_int32 * G)(*[int(x) for x in pos])
    op = Op("bl_attention_decode_rope_grouped_bf16", lib.bl_attention_decode_rope_grouped_bf16,
            (C.byref(d), cos.data_ptr(), sin.data_ptr(), G, kp, vp, pp), (d, qkv, o, cos, sin, kp, vp, pp, list(k_caches), list(v_caches)))
    if run:
        op.run()
    return op


def preprocess_u8(frames: torch.Tensor, mean_std: torch.Tensor, out: torch.Tensor, run: bool = True) -> Op:
    """uint8 frames [B, H, W, 3] already at the model resolution → pixel_values [B, 6, H, W] bf16 (bl_preprocess_u8_bf16)."""
    lib = _lib.load()
    if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous() or frames.dim() != 4 or frames.shape[-1] != 3:
        raise TypeError("preprocess_u8: frames must be a contiguous CUDA/HIP uint8 tensor [B, H, W, 3]")
    B, H, W, _ = frames.shape
    if tuple(out.shape) != (B, 6, H, W) or not out.is_contiguous() or mean_std.dtype != torch.float32 or mean_std.numel() != 12:
        raise ValueError("preprocess_u8: out must be contiguous [B, 6, H, W] bf16, mean_std 12 floats")
    op = Op("bl_preprocess_u8_bf16", lib.bl_preprocess_u8_bf16, (frames.data_ptr(), B, H, W, mean_std.data_ptr(), _bf16(out, "out").data_ptr()),
            (frames, mean_std, out), nbytes=B * H * W * (3 + 12.0))
    if run:
        op.run()
    return op


def _bicubic(x: float) -> float:
    """Pillow's bicubic kernel (a = -0.5, support 2)."""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _lanczos(x: float) -> float:
    """Pillow's Lanczos-3 kernel (Resample.c lanczos_filter: truncated sinc, support 3)."""
    def sinc(v: float) -> float:
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v
    return sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


@functools.lru_cache(maxsize=64)
def resample_coeffs(in_size: int, out_size: int, filter: str = "bicubic"):
    """Pillow's coefficient table for one axis of an 8-bit separable resize (libImaging/Resample.c precompute_coeffs +
    normalize_coeffs_8bpc, whole-image box) with the bicubic (a = -0.5, support 2) or Lanczos-3 filter: (bounds int32
    [out, 2] = first tap / tap count, coefs int32 [out, ksize], ksize). Doubles throughout, in Pillow's operation order;
    22-bit fixed point, rounded half away from zero."""
    kernel, base_support = _FILTERS[filter]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = torch.zeros(out_size, 2, dtype=torch.int32)
    coefs = torch.zeros(out_size, ksize, dtype=torch.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [kernel((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
        for x, v in enumerate(w):
            coefs[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
    return bounds, coefs, ksize


_COEF_DEV: dict = {}


def resize_u8(frames: torch.Tensor, out_h: int, out_w: int, filter: str = "bicubic") -> torch.Tensor:
    """uint8 frames [B, H, W, 3] on the GPU → [B, out_h, out_w, 3], bit-identical to PIL `Image.resize((out_w, out_h),
    BICUBIC | LANCZOS)` per frame: horizontal pass, uint8 intermediate, vertical pass (a pass whose size does not change
    is skipped, as in Pillow)."""
    lib = _lib.load()
    if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous() or frames.dim() != 4 or frames.shape[-1] != 3:
        raise TypeError("resize_u8: frames must be a contiguous CUDA/HIP uint8 tensor [B, H, W, 3]")
    B, H, W, _ = frames.shape
    cur = frames

    def tables(n_in, n_out):
        key = (n_in, n_out, filter, frames.device)
        if key not in _COEF_DEV:
            b, c, ks = resample_coeffs(n_in, n_out, filter)
            _COEF_DEV[key] = (b.to(frames.device), c.to(frames.device), ks)
        return _COEF_DEV[key]

    if W != out_w:
        b, c, ks = tables(W, out_w)
        dst = torch.empty(B, H, out_w, 3, dtype=torch.uint8, device=frames.device)
        Op("bl_resample_pass_u8", lib.bl_resample_pass_u8, (cur.data_ptr(), dst.data_ptr(), B, H, W, out_w, 1, b.data_ptr(),
                                                             c.data_ptr(), ks), (cur, dst, b, c)).run()
        cur = dst
    if H != out_h:
        b, c, ks = tables(H, out_h)
        dst = torch.empty(B, out_h, out_w, 3, dtype=torch.uint8, device=frames.device)
        Op("bl_resample_pass_u8", lib.bl_resample_pass_u8, (cur.data_ptr(), dst.data_ptr(), B, out_w, H, out_h, 0, b.data_ptr(),
                                                             c.data_ptr(), ks), (cur, dst, b, c)).run()
        cur = dst
    return cur


def resize_bicubic_u8(frames: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    return resize_u8(frames, out_h, out_w, "bicubic")


def crop_resize_bilinear_u8(frames: torch.Tensor, y_base: float, y_step: float, x_base: float, x_step: float,
                            out_h: int, out_w: int, out: Optional[torch.Tensor] = None, run: bool = True):
    """tf.image.crop_and_resize (+ the uint8 ↔ float32 conversions around it) of the eval-time centre crop
    (experiments/robot/openvla_utils.py:81-155) on uint8 frames [B, H, W, 3] in HBM; the four fp32 sampling constants
    come from the caller (vla/eval_preprocess.py computes them exactly as its host restatement does)."""
    lib = _lib.load()
    if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous() or frames.dim() != 4 or frames.shape[-1] != 3:
        raise TypeError("crop_resize_bilinear_u8: frames must be a contiguous CUDA/HIP uint8 tensor [B, H, W, 3]")
    B, H, W, _ = frames.shape
    if out is None:
        out = torch.empty(B, out_h, out_w, 3, dtype=torch.uint8, device=frames.device)
    elif tuple(out.shape) != (B, out_h, out_w, 3) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError(f"crop_resize_bilinear_u8: out must be contiguous uint8 [{B}, {out_h}, {out_w}, 3]")
    op = Op("bl_crop_resize_bilinear_u8", lib.bl_crop_resize_bilinear_u8,
            (frames.data_ptr(), out.data_ptr(), B, H, W, out_h, out_w, float(y_base), float(y_step), float(x_base), float(x_step)),
            (frames, out), nbytes=B * (H * W + out_h * out_w) * 3.0)
    if run:
        op.run()
        return out
    return op


def fill_synth(dst: torch.Tensor, seed: int, mean: float, scale: float, *, rows: Optional[int] = None,
               cols: Optional[int] = None, ld: Optional[int] = None, run: bool = True) -> Op:
    """Deterministic synthetic fill (see oracle/synth.py for the CPU restatement of the same generator)."""
    lib = _lib.load()
    _bf16(dst, "dst")
    if rows is None:
        rows, cols, ld = 1, dst.numel(), dst.numel()
    op = Op("bl_fill_synth_bf16_2d", lib.bl_fill_synth_bf16_2d,
            (dst.data_ptr(), rows, cols, ld, seed & 0xFFFFFFFF, float(mean), float(scale)), (dst,))
    if run:
        op.run()
    return op
