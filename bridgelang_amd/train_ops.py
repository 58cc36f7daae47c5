"""Torch-tensor front end for the training-step entry points of the C ABI (backward + optimizer kernels, train.hip).
Immediate-mode wrappers: each call enqueues its kernel(s) on the current stream."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .ops import _bf16, _rows, _stream


def _ck(rc: int, name: str) -> None:
    if rc != 0:
        _lib.check(rc, name)


def cross_entropy_backward(logits: torch.Tensor, targets: torch.Tensor, mean_and_count: torch.Tensor,
                           dlogits: torch.Tensor, ignore_index: int = -100) -> None:
    rows, n = logits.shape
    _ck(_lib.load().bl_cross_entropy_backward_f32(logits.data_ptr(), _rows(logits, "logits"), rows, n, targets.data_ptr(),
                                                  ignore_index, mean_and_count.data_ptr(), _bf16(dlogits, "dlogits").data_ptr(),
                                                  _rows(dlogits, "dlogits"), _stream()), "bl_cross_entropy_backward_f32")


def rmsnorm_backward(x, w, dy, dx, dw, ws, eps: float, dres: Optional[torch.Tensor] = None) -> None:
    rows, dim = x.shape
    _ck(_lib.load().bl_rmsnorm_backward_bf16(
        _bf16(x, "x").data_ptr(), _rows(x, "x"), _bf16(w, "w").data_ptr(), _bf16(dy, "dy").data_ptr(), _rows(dy, "dy"),
        dres.data_ptr() if dres is not None else None, _rows(dres, "dres") if dres is not None else 0,
        _bf16(dx, "dx").data_ptr(), _rows(dx, "dx"), dw.data_ptr(), ws.data_ptr(), ws.numel(), rows, dim, float(eps),
        _stream()), "bl_rmsnorm_backward_bf16")


def colsum(a, out, ws) -> None:
    rows, cols = a.shape
    _ck(_lib.load().bl_colsum_bf16(_bf16(a, "a").data_ptr(), _rows(a, "a"), rows, cols, out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _stream()), "bl_colsum_bf16")


def swiglu(gu, act) -> None:
    rows, two_i = gu.shape
    _ck(_lib.load().bl_swiglu_bf16(_bf16(gu, "gu").data_ptr(), _rows(gu, "gu"), _bf16(act, "act").data_ptr(), _rows(act, "act"),
                                   rows, two_i // 2, _stream()), "bl_swiglu_bf16")


def swiglu_backward(gu, dact, dgu) -> None:
    rows, two_i = gu.shape
    _ck(_lib.load().bl_swiglu_backward_bf16(_bf16(gu, "gu").data_ptr(), _rows(gu, "gu"), _bf16(dact, "dact").data_ptr(),
                                            _rows(dact, "dact"), _bf16(dgu, "dgu").data_ptr(), _rows(dgu, "dgu"), rows,
                                            two_i // 2, _stream()), "bl_swiglu_backward_bf16")


def gelu(x, y) -> None:
    assert x.is_contiguous() and y.is_contiguous()
    _ck(_lib.load().bl_gelu_bf16(_bf16(x, "x").data_ptr(), _bf16(y, "y").data_ptr(), x.numel(), _stream()), "bl_gelu_bf16")


def gelu_backward(x, dy, dx) -> None:
    assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous()
    _ck(_lib.load().bl_gelu_backward_bf16(_bf16(x, "x").data_ptr(), _bf16(dy, "dy").data_ptr(), _bf16(dx, "dx").data_ptr(),
                                          x.numel(), _stream()), "bl_gelu_backward_bf16")


def rope_backward(dqkv, cos, sin, *, B: int, S: int, H: int, head_dim: int, pos0: int = 0) -> None:
    assert dqkv.is_contiguous()
    _ck(_lib.load().bl_rope_backward_bf16(_bf16(dqkv, "dqkv").data_ptr(), B, S, H, head_dim, cos.data_ptr(), sin.data_ptr(),
                                          pos0, _stream()), "bl_rope_backward_bf16")


def transpose_pad(a, out, rows_pad: int) -> None:
    rows, cols = a.shape
    _ck(_lib.load().bl_transpose_pad_bf16(_bf16(a, "a").data_ptr(), _rows(a, "a"), rows, cols, _bf16(out, "out").data_ptr(),
                                          _rows(out, "out"), rows_pad, _stream()), "bl_transpose_pad_bf16")


def sumsq_partial(g: torch.Tensor, partial: torch.Tensor) -> None:
    _ck(_lib.load().bl_sumsq_partial_f32(g.data_ptr(), g.numel(), partial.data_ptr(), partial.numel(), _stream()),
        "bl_sumsq_partial_f32")


def clip_coef(partials: torch.Tensor, max_norm: float, out: torch.Tensor) -> None:
    _ck(_lib.load().bl_clip_coef_f32(partials.data_ptr(), partials.numel(), float(max_norm), out.data_ptr(), _stream()),
        "bl_clip_coef_f32")


def adamw(p, m, v, g, step: int, lr: float, *, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
          norm_coef: Optional[torch.Tensor] = None, p_bf16: Optional[torch.Tensor] = None) -> None:
    for t in (p, m, v, g):
        assert t.dtype == torch.float32 and t.is_contiguous()
    _ck(_lib.load().bl_adamw_f32(p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(),
                                 norm_coef.data_ptr() if norm_coef is not None else None, p.numel(), float(lr),
                                 float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                 p_bf16.data_ptr() if p_bf16 is not None else None, _stream()), "bl_adamw_f32")


def embed_backward(ids, dx, dw, n_patches: int) -> None:
    B, L = ids.shape
    _ck(_lib.load().bl_embed_backward_bf16(ids.data_ptr(), B, L, _bf16(dx, "dx").data_ptr(), dx.shape[-1], n_patches,
                                           dw.data_ptr(), _stream()), "bl_embed_backward_bf16")


def _adesc(q, k, v, o, B, H, Sq, Skv, head_dim, qs, ks, vs, os_, causal, scale, key_mask):
    from .ops import _attn_desc
    return _attn_desc(q, k, v, o, B, H, Sq, Skv, head_dim, qs, ks, vs, os_, causal,
                      head_dim ** -0.5 if scale is None else scale, key_mask)


def attention_lse(q, k, v, o, lse, *, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal,
                  scale=None, key_mask=None) -> None:
    """Training forward: attention + per-row base-2 log-sum-exp (lse: fp32 [B*H*pad32(Sq)])."""
    assert lse.dtype == torch.float32 and lse.numel() >= B * H * ((Sq + 31) // 32 * 32)
    d = _adesc(q, k, v, o, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal, scale, key_mask)
    import ctypes as C
    _ck(_lib.load().bl_attention_lse_bf16(C.byref(d), lse.data_ptr(), _stream()), "bl_attention_lse_bf16")


def attention_backward(q, k, v, o, dout, lse, delta, dq, dk, dv, *, B, H, Sq, Skv, head_dim, q_strides, k_strides,
                       v_strides, o_strides, causal, scale=None, key_mask=None) -> None:
    for t in (lse, delta):
        assert t.dtype == torch.float32 and t.numel() >= B * H * ((Sq + 31) // 32 * 32)
    d = _adesc(q, k, v, o, B, H, Sq, Skv, head_dim, q_strides, k_strides, v_strides, o_strides, causal, scale, key_mask)
    import ctypes as C
    _ck(_lib.load().bl_attention_backward_bf16(C.byref(d), _bf16(dout, "dout").data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                               _bf16(dq, "dq").data_ptr(), _bf16(dk, "dk").data_ptr(),
                                               _bf16(dv, "dv").data_ptr(), _stream()), "bl_attention_backward_bf16")
