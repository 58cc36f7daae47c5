"""TEST INFRASTRUCTURE — CPU restatement of the deterministic synthetic-tensor generator.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(bridgelang_amd/) never does.

Restates `synth_value` in bridgelang_amd/csrc/glue.hip bit for bit: a lowbias32 integer hash, the sum of four 16-bit
uniforms (Irwin-Hall, approximately normal), one exactly-rounded fp32 multiply and one exactly-rounded fp32 add, then
round-to-nearest-even to bf16. Because every step is integer arithmetic or a single IEEE operation, the GPU and the CPU
produce identical bits, so the oracle and the HIP path can hold the same "checkpoint" without ever copying weights.

There is no reference counterpart (the reference loads real checkpoints; none exists offline — SURVEY.md §8c); the
distribution mirrors the reference's init, normal(0, initializer_range): modeling_prismatic.py:185-205.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

IRWIN_HALL_SD = 37837.2265625
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 on uint64 lanes masked to 32 bits (numpy has no wrapping uint32 multiply without warnings)."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def tensor_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ ((seed * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF


def synth_f32(n: int, seed32: int, mean: float, std: float, start: int = 0) -> np.ndarray:
    """fp32 values (before the bf16 rounding) of logical elements start .. start+n-1."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    sm = _mix32(np.array([seed32], dtype=np.uint64))[0]
    h1 = _mix32(idx ^ sm)
    h2 = _mix32((h1 + np.uint64(0x9E3779B9)) & _M32)
    s = ((h1 & np.uint64(0xFFFF)) + (h1 >> np.uint64(16)) + (h2 & np.uint64(0xFFFF)) + (h2 >> np.uint64(16))).astype(np.int64)
    s = (s - 131070).astype(np.float32)
    scale = np.float32(std / IRWIN_HALL_SD)
    return np.float32(mean) + s * scale          # two separately rounded fp32 ops (no FMA in numpy)


_C = None


def _c_lib():
    """oracle/_ref/libbl_oracle.so (oracle/synth_c.c, built by oracle/Makefile) — the same generator in C for full-size
    checkpoints; None when it has not been built (the numpy path below is then used: identical bits, ~200x slower)."""
    global _C
    if _C is None:
        import ctypes
        from pathlib import Path
        so = Path(__file__).resolve().parent / "_ref" / "libbl_oracle.so"
        if so.exists():
            lib = ctypes.CDLL(str(so))
            lib.bl_oracle_synth_bf16.restype = None
            lib.bl_oracle_synth_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_int64]
            _C = lib
        else:
            _C = False
    return _C or None


def synth_bf16(shape: Tuple[int, ...], seed32: int, mean: float, std: float, use_c: bool = True) -> torch.Tensor:
    n = int(np.prod(shape))
    out = torch.empty(n, dtype=torch.bfloat16)
    lib = _c_lib() if use_c and n >= (1 << 16) else None
    if lib is not None:
        lib.bl_oracle_synth_bf16(out.data_ptr(), n, seed32 & 0xFFFFFFFF, float(np.float32(mean)),
                                 float(np.float32(std / IRWIN_HALL_SD)), 0)
        return out.view(shape)
    step = 1 << 24                                   # bound temporary memory for the 131 M-element embeddings
    for s in range(0, n, step):
        m = min(step, n - s)
        out[s:s + m] = torch.from_numpy(synth_f32(m, seed32, mean, std, s)).to(torch.bfloat16)   # RNE
    return out.view(shape)


def synth_state_dict(specs: Iterable, seed: int = 0, overlays: Iterable = ()) -> Dict[str, torch.Tensor]:
    """specs: objects with .name .shape .mean .std (bridgelang_amd.weights.tensor_specs) → HF-named bf16 CPU tensors.
    overlays (bridgelang_amd.weights.synthetic_overlays): row blocks generated on their own (.name seeds them) and written
    over rows [.row0, .row0 + .shape[0]) of tensor .base — the same two fills the device generator performs."""
    sd = {s.name: synth_bf16(tuple(s.shape), tensor_seed(s.name, seed), s.mean, s.std) for s in specs}
    for ov in overlays:
        if ov.base in sd:
            sd[ov.base][ov.row0:ov.row0 + ov.shape[0]] = synth_bf16(tuple(ov.shape), tensor_seed(ov.name, seed), ov.mean, ov.std)
    return sd


def dropout_keep(seed: int, salt: int, rows: int, cols: int, p: float) -> np.ndarray:
    """The keep mask [rows, cols] (bool) of bl_dropout_bf16 for device seed value `seed` and adapted linear `salt`:
    kept iff the top 24 bits of mix32((row·cols + col) ^ mix32(mix32(seed) + salt)) >= round(p · 2^24). Test infrastructure."""
    key = _mix32((_mix32(np.array([seed & 0xFFFFFFFF], dtype=np.uint64)) + np.uint64(salt & 0xFFFFFFFF)) & _M32)[0]
    idx = np.arange(rows * cols, dtype=np.uint64)
    h = _mix32(idx ^ key)
    thr = np.uint64(int(np.rint(np.float32(p) * np.float32(16777216.0))))
    return ((h >> np.uint64(8)) >= thr).reshape(rows, cols)
