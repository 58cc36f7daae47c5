"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU checkers for the 8-bit bicubic resize of
`PrismaticImageProcessor.apply_transform` (processing_prismatic.py:128-145 → torchvision `resize` on a PIL image →
Pillow libImaging/Resample.c).

Pinned: Pillow itself is installed here and on the GPU box, so `pil_resize` IS the reference's arithmetic, not a
restatement; `two_pass` restates one horizontal + one vertical 8bpc pass in numpy int32 (Resample.c
ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc: ss = 2^21 + Σ pixel·coef, clip8(ss >> 22), uint8
intermediate) so that a coefficient table can be checked against Pillow without a GPU.
"""
import numpy as np
from PIL import Image


def pil_resize(frames: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """frames uint8 [B, H, W, 3] → [B, out_h, out_w, 3] with PIL bicubic, frame by frame."""
    return np.stack([np.asarray(Image.fromarray(f).resize((out_w, out_h), Image.BICUBIC), dtype=np.uint8) for f in frames])


def one_pass(src: np.ndarray, bounds: np.ndarray, coefs: np.ndarray, axis: int) -> np.ndarray:
    """src uint8 [B, H, W, 3]; resample along axis 1 (vertical) or 2 (horizontal) with the given fixed-point table."""
    src = np.moveaxis(src, axis, -1).astype(np.int64)                  # [..., in_len]
    out = np.empty(src.shape[:-1] + (bounds.shape[0],), dtype=np.uint8)
    for o in range(bounds.shape[0]):
        first, n = int(bounds[o, 0]), int(bounds[o, 1])
        ss = (1 << 21) + (src[..., first:first + n] * coefs[o, :n].astype(np.int64)).sum(-1)
        out[..., o] = np.clip(ss >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def two_pass(frames: np.ndarray, tables_w, tables_h) -> np.ndarray:
    """tables_* = (bounds, coefs) or None when that axis keeps its size (Pillow skips the pass)."""
    cur = frames
    if tables_w is not None:
        cur = one_pass(cur, tables_w[0], tables_w[1], axis=2)
    if tables_h is not None:
        cur = one_pass(cur, tables_h[0], tables_h[1], axis=1)
    return cur
