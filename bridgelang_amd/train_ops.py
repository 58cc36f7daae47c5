"""Torch-tensor front end for the training-step entry points of the C ABI (train.hip, attention_bwd.hip).
Same convention as ops.py: every builder returns a prepared `Op` and (run=True) enqueues it on the current stream."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .ops import Op, _attn_desc, _bf16, _rows


def _op(name: str, args: tuple, keep: tuple, run: bool, nbytes: float = 0.0, flops: float = 0.0) -> Op:
    op = Op(name, getattr(_lib.load(), name), args, keep, flops=flops, nbytes=nbytes)
    if run:
        op.run()
    return op


def _f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{what}: expected a CUDA/HIP float32 tensor, got {t.dtype} on {t.device}")
    return t


def cross_entropy_backward(logits, targets, mean_and_count, dlogits, ignore_index: int = -100, run: bool = True) -> Op:
    rows, n = logits.shape
    return _op("bl_cross_entropy_backward_f32",
               (_f32(logits, "logits").data_ptr(), _rows(logits, "logits"), rows, n, targets.data_ptr(), ignore_index,
                mean_and_count.data_ptr(), _bf16(dlogits, "dlogits").data_ptr(), _rows(dlogits, "dlogits")),
               (logits, targets, mean_and_count, dlogits), run, nbytes=6.0 * rows * n)


def rmsnorm_backward(x, w, dy, dx, dw, ws, eps: float, dres: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    rows, dim = x.shape
    return _op("bl_rmsnorm_backward_bf16",
               (_bf16(x, "x").data_ptr(), _rows(x, "x"), _bf16(w, "w").data_ptr(), _bf16(dy, "dy").data_ptr(), _rows(dy, "dy"),
                dres.data_ptr() if dres is not None else None, _rows(dres, "dres") if dres is not None else 0,
                _bf16(dx, "dx").data_ptr(), _rows(dx, "dx"), _f32(dw, "dw").data_ptr(), _f32(ws, "ws").data_ptr(), ws.numel(),
                rows, dim, float(eps)), (x, w, dy, dx, dw, ws, dres), run, nbytes=2.0 * rows * dim * (4 if dres is not None else 3))


def colsum(a, out, ws, run: bool = True) -> Op:
    rows, cols = a.shape
    return _op("bl_colsum_bf16", (_bf16(a, "a").data_ptr(), _rows(a, "a"), rows, cols, _f32(out, "out").data_ptr(),
                                  _f32(ws, "ws").data_ptr(), ws.numel()), (a, out, ws), run, nbytes=2.0 * rows * cols)


def swiglu(gu, act, run: bool = True) -> Op:
    rows, two_i = gu.shape
    return _op("bl_swiglu_bf16", (_bf16(gu, "gu").data_ptr(), _rows(gu, "gu"), _bf16(act, "act").data_ptr(), _rows(act, "act"),
                                  rows, two_i // 2), (gu, act), run, nbytes=3.0 * rows * two_i)


def swiglu_backward(gu, dact, dgu, run: bool = True) -> Op:
    rows, two_i = gu.shape
    return _op("bl_swiglu_backward_bf16",
               (_bf16(gu, "gu").data_ptr(), _rows(gu, "gu"), _bf16(dact, "dact").data_ptr(), _rows(dact, "dact"),
                _bf16(dgu, "dgu").data_ptr(), _rows(dgu, "dgu"), rows, two_i // 2), (gu, dact, dgu), run, nbytes=5.0 * rows * two_i)


def gelu(x, y, run: bool = True) -> Op:
    rows, cols = x.shape
    return _op("bl_gelu_bf16", (_bf16(x, "x").data_ptr(), _rows(x, "x"), _bf16(y, "y").data_ptr(), _rows(y, "y"), rows, cols), (x, y), run,
               nbytes=4.0 * rows * cols)


def gelu_backward(x, dy, dx, run: bool = True) -> Op:
    rows, cols = x.shape
    return _op("bl_gelu_backward_bf16", (_bf16(x, "x").data_ptr(), _rows(x, "x"), _bf16(dy, "dy").data_ptr(), _rows(dy, "dy"),
                                         _bf16(dx, "dx").data_ptr(), _rows(dx, "dx"), rows, cols), (x, dy, dx), run, nbytes=6.0 * rows * cols)


def rope(qkv, cos, sin, *, B: int, S: int, H: int, head_dim: int, pos0: int = 0, run: bool = True) -> Op:
    """Rotate the q and k thirds of fused qkv rows [B*S, 3*H*hd] (leading dimension free) in place: no KV cache."""
    return _op("bl_rope_bf16", (_bf16(qkv, "qkv").data_ptr(), _rows(qkv, "qkv"), B, S, H, head_dim, cos.data_ptr(), sin.data_ptr(), pos0),
               (qkv, cos, sin), run, nbytes=8.0 * B * S * H * head_dim)


def rope_backward(dqkv, cos, sin, *, B: int, S: int, H: int, head_dim: int, pos0: int = 0, run: bool = True) -> Op:
    return _op("bl_rope_backward_bf16", (_bf16(dqkv, "dqkv").data_ptr(), _rows(dqkv, "dqkv"), B, S, H, head_dim, cos.data_ptr(),
                                         sin.data_ptr(), pos0), (dqkv, cos, sin), run, nbytes=8.0 * B * S * H * head_dim)


def transpose_pad(a, out, rows_pad: int, run: bool = True) -> Op:
    rows, cols = a.shape
    return _op("bl_transpose_pad_bf16", (_bf16(a, "a").data_ptr(), _rows(a, "a"), rows, cols, _bf16(out, "out").data_ptr(),
                                         _rows(out, "out"), rows_pad), (a, out), run, nbytes=2.0 * cols * (rows + rows_pad))


def colamax(x: torch.Tensor, amax: torch.Tensor, run: bool = True) -> Op:
    """amax[c] = max(amax[c], max_t |x[t, c]|) (fp32; zero it first): per-channel scales of the e4m3 weight-gradient operands."""
    rows, cols = x.shape
    assert amax.dtype == torch.float32 and amax.numel() >= cols
    return _op("bl_colamax_bf16", (_bf16(x, "x").data_ptr(), _rows(x, "x"), rows, cols, amax.data_ptr()), (x, amax), run,
               nbytes=2.0 * rows * cols)


def transpose_quantize_fp8(x: torch.Tensor, amax: torch.Tensor, q: torch.Tensor, scales: torch.Tensor, tokens_pad: int, packed: bool,
                           run: bool = True) -> Op:
    """x bf16 [tokens, channels] → e4m3 codes, token-contiguous per channel (tokens zero-padded to tokens_pad) + scales[c] =
    amax[c] / 448: q uint8 [channels, tokens_pad] row-major, or (packed) in bl_gemm_fp8's weight-operand packing
    [channels / 16, tokens_pad / 64, 64, 16]. One pass: 2 B read + 1 B written per element."""
    rows, cols = x.shape
    assert q.dtype == torch.uint8 and q.is_contiguous() and q.numel() == cols * tokens_pad and scales.numel() >= cols
    return _op("bl_transpose_quantize_fp8", (_bf16(x, "x").data_ptr(), _rows(x, "x"), rows, cols, amax.data_ptr(), q.data_ptr(),
                                             tokens_pad, int(packed), scales.data_ptr()), (x, amax, q, scales), run,
               nbytes=3.0 * rows * cols)


def map_rows(src, dst, *, rows: int, group: int, stride: int, offset: int, scatter: bool, run: bool = True) -> Op:
    cols = src.shape[1]
    return _op("bl_map_rows_bf16", (_bf16(src, "src").data_ptr(), _rows(src, "src"), _bf16(dst, "dst").data_ptr(), _rows(dst, "dst"),
                                    rows, cols, group, stride, offset, int(scatter)), (src, dst), run, nbytes=4.0 * rows * cols)


def sumsq_partial(g: torch.Tensor, partial: torch.Tensor, run: bool = True) -> Op:
    return _op("bl_sumsq_partial_f32", (_f32(g, "g").data_ptr(), g.numel(), _f32(partial, "partial").data_ptr(), partial.numel()),
               (g, partial), run, nbytes=4.0 * g.numel())


def clip_coef(partials: torch.Tensor, max_norm: float, out: torch.Tensor, run: bool = True) -> Op:
    return _op("bl_clip_coef_f32", (partials.data_ptr(), partials.numel(), float(max_norm), out.data_ptr()), (partials, out), run)


def adamw(p, m, v, g, step: int, lr: float, *, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
          norm_coef: Optional[torch.Tensor] = None, p_bf16: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    for t in (p, m, v, g):
        assert t.dtype == torch.float32 and t.is_contiguous()
    return _op("bl_adamw_f32", (p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(),
                                norm_coef.data_ptr() if norm_coef is not None else None, p.numel(), float(lr), float(betas[0]),
                                float(betas[1]), float(eps), float(weight_decay), int(step),
                                p_bf16.data_ptr() if p_bf16 is not None else None), (p, m, v, g, norm_coef, p_bf16), run,
               nbytes=(28.0 + (2.0 if p_bf16 is not None else 0.0)) * p.numel())


def embed_backward(ids, dx, dw, n_patches: int, run: bool = True) -> Op:
    B, L = ids.shape
    return _op("bl_embed_backward_bf16", (ids.data_ptr(), B, L, _bf16(dx, "dx").data_ptr(), dx.shape[-1], n_patches,
                                          _f32(dw, "dw").data_ptr()), (ids, dx, dw), run)


def attention_lse(q, k, v, o, lse, *, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal,
                  scale=None, key_mask=None, run: bool = True) -> Op:
    """Training forward: attention + per-row base-2 log-sum-exp (lse: fp32 [B*H*pad32(Sq)])."""
    assert _f32(lse, "lse").numel() >= B * H * ((Sq + 31) // 32 * 32)
    d = _attn_desc(q, k, v, o, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal,
                   head_dim ** -0.5 if scale is None else scale, key_mask)
    return _op("bl_attention_lse_bf16", (C.byref(d), lse.data_ptr()), (d, q, k, v, o, lse, key_mask), run,
               flops=4.0 * B * H * Sq * Skv * head_dim * (0.5 if causal else 1.0))


def attention_backward(q, k, v, o, dout, lse, delta, dq, dk, dv, *, B, H, Sq, Skv, head_dim, q_strides, k_strides,
                       v_strides, o_strides, causal, scale=None, key_mask=None, run: bool = True) -> Op:
    for t in (lse, delta):
        assert _f32(t, "lse/delta").numel() >= B * H * ((Sq + 31) // 32 * 32)
    d = _attn_desc(q, k, v, o, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal,
                   head_dim ** -0.5 if scale is None else scale, key_mask)
    return _op("bl_attention_backward_bf16",
               (C.byref(d), _bf16(dout, "dout").data_ptr(), lse.data_ptr(), delta.data_ptr(), _bf16(dq, "dq").data_ptr(),
                _bf16(dk, "dk").data_ptr(), _bf16(dv, "dv").data_ptr()), (d, q, k, v, o, dout, lse, delta, dq, dk, dv, key_mask), run,
               flops=14.0 * B * H * Sq * Skv * head_dim * (0.5 if causal else 1.0))


def pack(w: torch.Tensor, out: torch.Tensor, run: bool = True) -> Op:
    """Row-major [N, K] → fragment-major packed (bl_pack_weight_bf16) as a replayable Op (weights after an optimizer
    step; the transposed activations of a wgrad GEMM)."""
    N, K = w.shape
    assert out.numel() == N * K and N % 16 == 0 and K % 32 == 0
    return _op("bl_pack_weight_bf16", (_bf16(w, "w").data_ptr(), _rows(w, "w"), N, K, _bf16(out, "out").data_ptr()), (w, out), run,
               nbytes=4.0 * N * K)


def layernorm_backward(x, w, dy, dx, dw, db, ws, eps: float, dres: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    rows, dim = x.shape
    return _op("bl_layernorm_backward_bf16",
               (_bf16(x, "x").data_ptr(), _rows(x, "x"), _bf16(w, "w").data_ptr(), _bf16(dy, "dy").data_ptr(), _rows(dy, "dy"),
                dres.data_ptr() if dres is not None else None, _rows(dres, "dres") if dres is not None else 0,
                _bf16(dx, "dx").data_ptr(), _rows(dx, "dx"), _f32(dw, "dw").data_ptr(), _f32(db, "db").data_ptr(),
                _f32(ws, "ws").data_ptr(), ws.numel(), rows, dim, float(eps)), (x, w, dy, dx, dw, db, ws, dres), run,
               nbytes=2.0 * rows * dim * (4 if dres is not None else 3))


def scale_residual(u, scale, res, y, run: bool = True) -> Op:
    rows, cols = u.shape
    return _op("bl_scale_residual_bf16", (_bf16(u, "u").data_ptr(), _rows(u, "u"), _bf16(scale, "scale").data_ptr(),
                                          _bf16(res, "res").data_ptr(), _rows(res, "res"), _bf16(y, "y").data_ptr(), _rows(y, "y"),
                                          rows, cols), (u, scale, res, y), run, nbytes=6.0 * rows * cols)


def layerscale_backward(dy, u, scale, du, dscale, ws, run: bool = True) -> Op:
    rows, cols = dy.shape
    return _op("bl_layerscale_backward_bf16",
               (_bf16(dy, "dy").data_ptr(), _rows(dy, "dy"), _bf16(u, "u").data_ptr(), _rows(u, "u"), _bf16(scale, "scale").data_ptr(),
                _bf16(du, "du").data_ptr(), _rows(du, "du"), _f32(dscale, "dscale").data_ptr(), _f32(ws, "ws").data_ptr(), ws.numel(),
                rows, cols), (dy, u, scale, du, dscale, ws), run, nbytes=6.0 * rows * cols)


def fill_zero(t: torch.Tensor, run: bool = True) -> Op:
    assert t.is_contiguous()
    return _op("bl_memset_zero", (t.data_ptr(), t.numel() * t.element_size()), (t,), run, nbytes=float(t.numel() * t.element_size()))


def copy_f32(src: torch.Tensor, dst: torch.Tensor, run: bool = True) -> Op:
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel() and src.dtype == dst.dtype
    return _op("bl_copy_bytes", (dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size()), (src, dst), run)


def scale(x: torch.Tensor, s: float, out: torch.Tensor, run: bool = True) -> Op:
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel()
    return _op("bl_scale_bf16", (_bf16(x, "x").data_ptr(), float(s), _bf16(out, "out").data_ptr(), x.numel()), (x, out), run)


def dropout(x: torch.Tensor, out: torch.Tensor, p: float, seed: torch.Tensor, salt: int, run: bool = True) -> Op:
    """out = nn.Dropout(p)(x) in training mode with the counter-based mask of (seed tensor [1] uint32 on the device, salt)."""
    rows, cols = x.shape
    assert seed.dtype in (torch.int32, torch.uint32) and seed.is_cuda and seed.numel() == 1
    return _op("bl_dropout_bf16", (_bf16(x, "x").data_ptr(), _rows(x, "x"), rows, cols, float(p), seed.data_ptr(), int(salt) & 0xFFFFFFFF,
                                   _bf16(out, "out").data_ptr(), _rows(out, "out")), (x, out, seed), run, nbytes=4.0 * rows * cols)


def dropout_grad_fix(u: torch.Tensor, dx: torch.Tensor, p: float, seed: torch.Tensor, salt: int, run: bool = True) -> Op:
    """dx (= dy·W + u) → dy·W + mask/(1-p) ⊙ u with the mask of `dropout(…, seed, salt)`."""
    rows, cols = u.shape
    return _op("bl_dropout_grad_fix_bf16", (_bf16(u, "u").data_ptr(), _rows(u, "u"), rows, cols, float(p), seed.data_ptr(),
                                            int(salt) & 0xFFFFFFFF, _bf16(dx, "dx").data_ptr(), _rows(dx, "dx")), (u, dx, seed), run,
               nbytes=6.0 * rows * cols)


def lora_block_mask(g: torch.Tensor, rp: int, members: int, interleave: bool, run: bool = True) -> Op:
    n, R = g.shape
    assert g.is_contiguous()
    return _op("bl_lora_block_mask_f32", (_f32(g, "g").data_ptr(), n, R, rp, members, int(interleave)), (g,), run)


def transpose_pack(a: torch.Tensor, out_packed: torch.Tensor, rows_pad: int, run: bool = True) -> Op:
    """[rows, cols] → packed [cols/16, rows_pad/32, 64, 8] (the transposed matrix in fragment-major layout)."""
    rows, cols = a.shape
    assert out_packed.numel() == cols * rows_pad and out_packed.is_contiguous()
    return _op("bl_transpose_pack_bf16", (_bf16(a, "a").data_ptr(), _rows(a, "a"), rows, cols, _bf16(out_packed, "out").data_ptr(),
                                          rows_pad), (a, out_packed), run, nbytes=2.0 * cols * (rows + rows_pad))


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor, workspace: Optional[torch.Tensor] = None, run: bool = True) -> Op:
    """out[N, K] (fp32) = dyᵀ·x over the token rows: dy [T, N], x [T, K] row-major bf16 (column-slice views allowed) — the
    weight gradient of y = x·Wᵀ read straight from the buffers the backward pass already holds (bl_gemm_tn_bf16)."""
    from .ops import EPI_F32, GemmDesc
    Tn, N = dy.shape
    K = x.shape[1]
    if x.shape[0] != Tn or tuple(out.shape) != (N, K):
        raise ValueError(f"gemm_tn: dy{tuple(dy.shape)} x{tuple(x.shape)} out{tuple(out.shape)}")
    d = GemmDesc()
    d.A, d.lda = _bf16(dy, "dy").data_ptr(), _rows(dy, "dy")
    d.W, d.ldw = _bf16(x, "x").data_ptr(), _rows(x, "x")
    d.C, d.ldc = _f32(out, "out").data_ptr(), _rows(out, "out")
    d.M, d.N, d.K, d.epilogue = N, K, Tn, EPI_F32
    keep = [d, dy, x, out]
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
        keep.append(workspace)
    return _op("bl_gemm_tn_bf16", (C.byref(d),), tuple(keep), run, flops=2.0 * Tn * N * K, nbytes=2.0 * Tn * (N + K) + 4.0 * N * K)


def gemm_tn_small(P: torch.Tensor, Q: torch.Tensor, C: torch.Tensor, transpose_out: bool, ws: Optional[torch.Tensor] = None,
                  alpha: float = 1.0, run: bool = True) -> Op:
    """C = alpha · Pᵀ·Q over the rows: P [T, R] (R in 64/128/192), Q [T, N]; C fp32 [R, N] or, transposed, [N, R]."""
    Tn, R = P.shape
    N = Q.shape[1]
    assert Q.shape[0] == Tn and tuple(C.shape) == ((N, R) if transpose_out else (R, N)) and C.is_contiguous()
    return _op("bl_gemm_tn_small_bf16",
               (_bf16(P, "P").data_ptr(), _rows(P, "P"), _bf16(Q, "Q").data_ptr(), _rows(Q, "Q"), Tn, R, N, _f32(C, "C").data_ptr(),
                C.shape[1], int(transpose_out), float(alpha), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0),
               (P, Q, C, ws), run, flops=2.0 * Tn * R * N, nbytes=2.0 * Tn * (R + N) + 4.0 * R * N)


def pack_into(w: torch.Tensor, out: torch.Tensor, kt_total: int, kb_offset: int, run: bool = True) -> Op:
    """Row-major [N, K] → k-blocks [kb_offset, kb_offset + K/32) of the packed matrix `out` ([N/16, kt_total, 64, 8])."""
    N, K = w.shape
    assert out.numel() == N * kt_total * 32
    return _op("bl_pack_weight_into_bf16", (_bf16(w, "w").data_ptr(), _rows(w, "w"), N, K, _bf16(out, "out").data_ptr(), kt_total,
                                            kb_offset), (w, out), run, nbytes=4.0 * N * K)


def transpose_pack_into(a: torch.Tensor, out: torch.Tensor, rows_pad: int, kt_total: int, kb_offset: int, run: bool = True) -> Op:
    """[rows, cols] → the transposed matrix's k-blocks [kb_offset, …) of packed `out` ([cols/16, kt_total, 64, 8])."""
    rows, cols = a.shape
    assert out.numel() == cols * kt_total * 32
    return _op("bl_transpose_pack_into_bf16", (_bf16(a, "a").data_ptr(), _rows(a, "a"), rows, cols, _bf16(out, "out").data_ptr(),
                                               rows_pad, kt_total, kb_offset), (a, out), run, nbytes=2.0 * cols * (rows + rows_pad))


# ---- batched small ops: one launch for a table of independent copies / (scaled) packs / (scaled) transposing packs -------------
def _grid_for(total: int, block: int = 256) -> int:
    return max(1, min(2048, (total + block - 1) // block))


def be_copy(src: torch.Tensor, dst: torch.Tensor):
    """Table entry: dst ← src (bytes), the work of bl_copy_bytes."""
    from ._lib import BatchOpDesc
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() * src.element_size() == dst.numel() * dst.element_size()
    n = src.numel() * src.element_size()
    vec = ((src.data_ptr() | dst.data_ptr()) & 15) == 0
    d = BatchOpDesc(kind=0, nblocks=_grid_for((n + 15) // 16 if vec else n), n=n, src=src.data_ptr(), dst=dst.data_ptr(), scale=1.0)
    return d, (src, dst)


def be_pack(w: torch.Tensor, out: torch.Tensor, kt_total: Optional[int] = None, kb_offset: int = 0, scale: float = 1.0):
    """Table entry: row-major [N, K] → k-blocks [kb_offset, …) of packed `out` (bl_pack_weight(_into)_bf16), of bf16(scale · w)."""
    from ._lib import BatchOpDesc
    N, K = w.shape
    kt = K // 32 if kt_total is None else kt_total
    assert N % 16 == 0 and K % 32 == 0 and out.numel() == N * kt * 32 and kb_offset + K // 32 <= kt and w.stride(1) == 1
    d = BatchOpDesc(kind=1, nblocks=_grid_for(N * K // 8), rows=N, cols=K, ld=w.stride(0), kt_total=kt, kb_off=kb_offset,
                    src=_bf16(w, "w").data_ptr(), dst=_bf16(out, "out").data_ptr(), scale=float(scale))
    return d, (w, out)


def be_transpose_pack(a: torch.Tensor, out: torch.Tensor, rows_pad: int, kt_total: Optional[int] = None, kb_offset: int = 0,
                      scale: float = 1.0):
    """Table entry: [rows, cols] → the transposed matrix's k-blocks of packed `out` (bl_transpose_pack(_into)_bf16), of bf16(scale · a)."""
    from ._lib import BatchOpDesc
    rows, cols = a.shape
    kt = rows_pad // 32 if kt_total is None else kt_total
    assert cols % 64 == 0 and rows_pad % 32 == 0 and rows_pad >= rows and out.numel() == cols * kt * 32 and a.stride(1) == 1
    bx = cols // 64
    d = BatchOpDesc(kind=2, nblocks=bx * ((rows_pad + 255) // 256), rows=rows, cols=cols, rows_pad=rows_pad, bx_count=bx,
                    ld=a.stride(0), kt_total=kt, kb_off=kb_offset, src=_bf16(a, "a").data_ptr(), dst=_bf16(out, "out").data_ptr(),
                    scale=float(scale))
    return d, (a, out)


def batched(entries, device, run: bool = True) -> Op:
    """ONE launch (bl_batched_ops) for the table entries made by be_copy / be_pack / be_transpose_pack. The entries must be
    independent of each other (no entry reads what another one writes)."""
    import ctypes as C
    import numpy as np
    from ._lib import BatchOpDesc
    descs = [e[0] for e in entries]
    arr = (BatchOpDesc * len(descs))(*descs)
    table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)
    starts, tot = [], 0
    for d in descs:
        starts.append(tot)
        tot += d.nblocks
    block_start = torch.tensor(starts, dtype=torch.int32, device=device)
    keep = tuple(t for e in entries for t in e[1])
    nbytes = float(sum((d.n if d.kind == 0 else 2 * d.rows * d.cols) * 2 for d in descs))
    return _op("bl_batched_ops", (table.data_ptr(), block_start.data_ptr(), len(descs), tot), (table, block_start) + keep, run, nbytes=nbytes)


def cast(src: torch.Tensor, dst: torch.Tensor, run: bool = True) -> Op:
    """fp32 → bf16 (round to nearest even) or bf16 → fp32 of flat contiguous buffers (gradient wire format)."""
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    name = {(torch.float32, torch.bfloat16): "bl_cast_f32_bf16", (torch.bfloat16, torch.float32): "bl_cast_bf16_f32"}[(src.dtype, dst.dtype)]
    return _op(name, (src.data_ptr(), dst.data_ptr(), src.numel()), (src, dst), run, nbytes=6.0 * src.numel())


def axpy(y: torch.Tensor, x: torch.Tensor, a: float, run: bool = True) -> Op:
    """y += a·x (flat fp32): gradient accumulation."""
    assert y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel()
    return _op("bl_axpy_f32", (_f32(y, "y").data_ptr(), _f32(x, "x").data_ptr(), float(a), y.numel()), (y, x), run, nbytes=12.0 * y.numel())
