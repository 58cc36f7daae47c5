"""Eval-time image preprocessing of the robot evaluation loops, host side:

  * `center_crop_and_resize` — `get_vla_action(..., center_crop=True)` (experiments/robot/openvla_utils.py:81-155): crop the
    centre box of area `crop_scale` × image area (side × sqrt(crop_scale)) and resize it back with
    `tf.image.crop_and_resize` semantics (bilinear samples at box corners inclusive, (out-1) intervals), through
    float32 in [0, 1] and back to uint8 with TF's saturating `convert_image_dtype` (× 255.5, truncate).
  * `resize_image` — LIBERO's `resize_image` (experiments/robot/libero/libero_utils.py:33-47): JPEG encode → decode
    (as the RLDS dataset builder stores frames), Lanczos-3 antialiased resize, round, clip to uint8.

The reference runs both through TensorFlow, which is absent here and on the GPU box: the arithmetic below restates the
TF ops' definitions with numpy / Pillow — the centre crop in the fp32 operation order of TF's own CPU kernel (box arithmetic
of openvla_utils.py:101-115 on float32, `top + (bottom − top)·lerp` interpolation), so the only gap left to the reference
pipeline is whether TF's build contracts multiply-adds; the JPEG codec and Lanczos filter are Pillow's (TF's differ in
rounding details) — PARITY UNPINNED against TF; the tests check the defining properties (geometry, identity cases, ranges).
Frames then go through `PrismaticImageProcessor` (bit-exact vs Pillow, on the CPU or on the GPU).

Device twins for uint8 frames already in HBM (SURVEY §8(f)2 "as a GPU kernel (K1) fed by uint8 frames"):
`center_crop_and_resize_gpu` (bl_crop_resize_bilinear_u8, bit-identical to `center_crop_and_resize` below) and
`lanczos_resize_gpu` (bl_resample_pass_u8 with Lanczos-3 tables, bit-identical to Pillow's `Image.resize(LANCZOS)`, the
resize half of `resize_image`). The JPEG round trip is an entropy codec, not a throughput kernel: it stays on the host."""
from __future__ import annotations

import io
from typing import Tuple

import numpy as np
from PIL import Image


def _to_float(img_u8: np.ndarray) -> np.ndarray:
    """tf.image.convert_image_dtype(uint8 → float32): x / 255."""
    return img_u8.astype(np.float32) * np.float32(1.0 / 255.0)


def _to_uint8_saturate(img_f: np.ndarray) -> np.ndarray:
    """tf.image.convert_image_dtype(float32 → uint8, saturate=True): x * (255 + 0.5), saturate, truncate."""
    return np.clip(img_f * np.float32(255.5), 0.0, 255.0).astype(np.uint8)


def sampling_constants(box: Tuple[float, float, float, float], hw: Tuple[int, int], out_hw: Tuple[int, int]):
    """The four fp32 constants of the sampling grid of TF's CropAndResize kernel: in_y = y1·(H-1) + i·height_scale with
    height_scale = (y2 - y1)·(H-1) / (out_h - 1), every operation in fp32 in TF's order (crop_and_resize_op.cc); a single
    output row / column samples the box centre."""
    (H, W), (oh, ow), (y1, x1, y2, x2) = hw, out_hw, (np.float32(v) for v in box)
    f32 = np.float32
    if oh == 1:
        y_base, y_step = f32(0.5) * (y1 + y2) * f32(H - 1), f32(0.0)
    else:
        y_base, y_step = y1 * f32(H - 1), (y2 - y1) * f32(H - 1) / f32(oh - 1)
    if ow == 1:
        x_base, x_step = f32(0.5) * (x1 + x2) * f32(W - 1), f32(0.0)
    else:
        x_base, x_step = x1 * f32(W - 1), (x2 - x1) * f32(W - 1) / f32(ow - 1)
    return float(y_base), float(y_step), float(x_base), float(x_step)


def center_crop_box(crop_scale: float) -> Tuple[float, float, float, float]:
    """openvla_utils.py:101-115 in the reference's own precision: tf.sqrt / clip / (1 - s) / 2 / offset + s on float32 tensors."""
    side = np.clip(np.sqrt(np.float32(crop_scale)), np.float32(0.0), np.float32(1.0))
    off = (np.float32(1.0) - side) / np.float32(2.0)
    return (float(off), float(off), float(off + side), float(off + side))


def center_crop_and_resize_gpu(frames_u8, crop_scale: float = 0.9, out_hw: Tuple[int, int] = (224, 224)):
    """`center_crop_and_resize` for a batch of uint8 frames [B, H, W, 3] resident on the GPU → uint8 [B, out_h, out_w, 3]
    (one HBM-bound kernel, bl_crop_resize_bilinear_u8); bit-identical to the host function below, frame by frame."""
    from .. import ops
    B, H, W, _ = frames_u8.shape
    yb, ys, xb, xs = sampling_constants(center_crop_box(crop_scale), (H, W), out_hw)
    return ops.crop_resize_bilinear_u8(frames_u8, yb, ys, xb, xs, out_hw[0], out_hw[1])


def lanczos_resize_gpu(frames_u8, resize_size: Tuple[int, int]):
    """The resize half of `resize_image` for uint8 frames [B, H, W, 3] on the GPU: Pillow's 8-bit Lanczos-3 resample as
    two bl_resample_pass_u8 passes, bit-identical to `Image.resize((w, h), Image.LANCZOS)`."""
    from .. import ops
    h, w = resize_size
    return ops.resize_u8(frames_u8, h, w, "lanczos")


def crop_and_resize_bilinear(img_f: np.ndarray, box: Tuple[float, float, float, float], out_hw: Tuple[int, int]) -> np.ndarray:
    """tf.image.crop_and_resize for one image [H, W, C] float32 and one normalised box (y1, x1, y2, x2), in the arithmetic of
    TF's CPU kernel (crop_and_resize_op.cc): in_y = y_base + i·y_step (fp32), top / bottom = floor / ceil, lerp = in − floor,
    top = tl + (tr − tl)·x_lerp, bottom = bl + (br − bl)·x_lerp, out = top + (bottom − top)·y_lerp — every operation rounded
    to fp32 on its own; 0 (the extrapolation value) outside the image."""
    H, W, _ = img_f.shape
    oh, ow = out_hw
    yb, ystep, xb, xstep = (np.float32(v) for v in sampling_constants(box, (H, W), out_hw))
    ys = yb + np.arange(oh, dtype=np.float32) * ystep
    xs = xb + np.arange(ow, dtype=np.float32) * xstep
    inside = ((ys >= 0) & (ys <= H - 1))[:, None, None] & ((xs >= 0) & (xs <= W - 1))[None, :, None]
    yt, yl = np.floor(ys), np.floor(xs)
    wy, wx = (ys - yt).astype(np.float32)[:, None, None], (xs - yl).astype(np.float32)[None, :, None]
    y0c, y1c = np.clip(yt.astype(np.int64), 0, H - 1), np.clip(np.ceil(ys).astype(np.int64), 0, H - 1)
    x0c, x1c = np.clip(yl.astype(np.int64), 0, W - 1), np.clip(np.ceil(xs).astype(np.int64), 0, W - 1)
    tl, tr, bl, br = img_f[y0c][:, x0c], img_f[y0c][:, x1c], img_f[y1c][:, x0c], img_f[y1c][:, x1c]
    top = tl + (tr - tl) * wx
    bot = bl + (br - bl) * wx
    return np.where(inside, top + (bot - top) * wy, np.float32(0.0)).astype(np.float32)


def center_crop_and_resize(image_u8: np.ndarray, crop_scale: float = 0.9, out_hw: Tuple[int, int] = (224, 224)) -> np.ndarray:
    """openvla_utils.py:127-155: uint8 [H, W, 3] → uint8 [out_h, out_w, 3]; the crop keeps sqrt(crop_scale) of each side."""
    out = crop_and_resize_bilinear(_to_float(np.asarray(image_u8)), center_crop_box(crop_scale), out_hw)
    return _to_uint8_saturate(np.clip(out, 0.0, 1.0))


def jpeg_round_trip(image_u8: np.ndarray, quality: int = 95) -> np.ndarray:
    """tf.image.encode_jpeg (defaults: quality 95, chroma down-sampling) followed by decode."""
    buf = io.BytesIO()
    Image.fromarray(np.asarray(image_u8, dtype=np.uint8)).save(buf, format="JPEG", quality=quality, subsampling="4:2:0")
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))


def resize_image(image_u8: np.ndarray, resize_size: Tuple[int, int]) -> np.ndarray:
    """libero_utils.py:33-47: JPEG round trip, Lanczos-3 antialiased resize to (height, width), round + clip to uint8."""
    assert isinstance(resize_size, tuple)
    img = jpeg_round_trip(image_u8)
    h, w = resize_size
    return np.asarray(Image.fromarray(img).resize((w, h), Image.LANCZOS), dtype=np.uint8)


# ---- the evaluation loops' policy call (experiments/robot/openvla_utils.py:115-172) --------------------------------------
OPENVLA_V01_SYSTEM_PROMPT = ("A chat between a curious user and an artificial intelligence assistant. "
                             "The assistant gives helpful, detailed, and polite answers to the user's questions.")


def vla_prompt(base_vla_name: str, task_label: str) -> str:
    """openvla_utils.py:157-163: the v0.1 chat form or the OpenVLA "In: … Out:" form."""
    if "openvla-v01" in base_vla_name:
        return f"{OPENVLA_V01_SYSTEM_PROMPT} USER: What action should the robot take to {task_label.lower()}? ASSISTANT:"
    return f"In: What action should the robot take to {task_label.lower()}?\nOut:"


def get_vla_action(vla, processor, base_vla_name: str, obs: dict, task_label: str, unnorm_key, center_crop: bool = False,
                   on_device: bool = True):
    """Drop-in for `get_vla_action` of the robot evaluation loops (same arguments, same returned action). `obs["full_image"]`
    is a uint8 RGB frame [H, W, 3] (numpy, or a torch tensor already in HBM). With `on_device` (default) the frame goes to
    the GPU once as uint8 and everything after runs there — centre crop + resize (bl_crop_resize_bilinear_u8), the image
    processor's bicubic resize + dual normalisation (bl_resample_pass_u8, bl_preprocess_u8_bf16), predict_action — each
    stage bit-identical to the host path (`on_device=False`: numpy crop → PIL → `processor(prompt, image)` as in the
    reference), so both give the same action."""
    import torch
    from PIL import Image
    prompt = vla_prompt(base_vla_name, task_label)
    frame = obs["full_image"]
    if not on_device:
        image = np.asarray(frame.cpu() if torch.is_tensor(frame) else frame, dtype=np.uint8)
        if center_crop:
            image = center_crop_and_resize(image, 0.9, (224, 224))
        inputs = processor(prompt, Image.fromarray(image).convert("RGB")).to(vla.device, dtype=torch.bfloat16)
        return vla.predict_action(**inputs, unnorm_key=unnorm_key, do_sample=False)
    dev = vla.device
    frames = (frame if torch.is_tensor(frame) else torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))).to(dev)
    frames = frames.view(1, *frames.shape[-3:]).contiguous()
    if center_crop:
        frames = center_crop_and_resize_gpu(frames, 0.9, (224, 224))
    pixel_values = processor.image_processor.preprocess_frames_gpu(frames)
    enc = processor.tokenizer(prompt, return_tensors="pt")
    return vla.predict_action(input_ids=enc["input_ids"].to(dev), attention_mask=enc["attention_mask"].to(dev),
                              pixel_values=pixel_values, unnorm_key=unnorm_key, do_sample=False)
