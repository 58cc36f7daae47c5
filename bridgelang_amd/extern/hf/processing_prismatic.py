"""Host-side preprocessing for the fused DINOv2 + SigLIP backbone: PIL image → `pixel_values [6, 224, 224]`.

Mirrors `PrismaticImageProcessor.apply_transform` / `PrismaticProcessor.__call__`
(prismatic/extern/hf/processing_prismatic.py:128-145,187-216) for the "resize-naive" strategy OpenVLA uses: per backbone
resize straight to 224×224 with PIL bicubic (what torchvision's functional `resize` does for PIL inputs), centre crop
(identity at 224), `to_tensor` (uint8 / 255, CHW fp32), normalise with the backbone's own mean/std (DINOv2: ImageNet;
SigLIP: 0.5/0.5 — timm data_cfg values recorded at convert_openvla_weights_to_hf.py:193-197), stack on the channel axis.
`preprocess` runs on the CPU exactly like the reference (a few hundred µs per frame); `preprocess_frames_gpu` is the
same arithmetic on the device for uint8 frames already in HBM.
torchvision is not installed here, so bit-exactness against it is unpinned; the arithmetic is the same torch ops.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
HALF = (0.5, 0.5, 0.5)


class PrismaticImageProcessor:
    model_input_names = ["pixel_values"]

    def __init__(self, use_fused_vision_backbone: bool = True, image_resize_strategy: str = "resize-naive",
                 input_sizes: Optional[List[Tuple[int, int, int]]] = None,
                 means: Optional[Sequence[Sequence[float]]] = None, stds: Optional[Sequence[Sequence[float]]] = None,
                 **_: Any) -> None:
        if image_resize_strategy != "resize-naive":
            raise ValueError(f"Image resize strategy `{image_resize_strategy}` is not supported on this path")
        self.use_fused_vision_backbone, self.image_resize_strategy = use_fused_vision_backbone, image_resize_strategy
        n = 2 if use_fused_vision_backbone else 1
        self.input_sizes = input_sizes if input_sizes is not None else [(3, 224, 224)] * n
        self.means = [tuple(m) for m in (means if means is not None else [IMAGENET_MEAN, HALF][:n])]
        self.stds = [tuple(s) for s in (stds if stds is not None else [IMAGENET_STD, HALF][:n])]

    def apply_transform(self, img: Image.Image) -> torch.Tensor:
        planes = []
        for (_, h, w), mean, std in zip(self.input_sizes, self.means, self.stds):
            im = img.resize((w, h), Image.BICUBIC)                                    # TVF.resize on a PIL image
            t = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous()
            t = t.to(torch.float32).div(255)                                          # TVF.to_tensor
            m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
            s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
            planes.append(t.sub_(m).div_(s))                                          # TVF.normalize
        return torch.vstack(planes)

    def preprocess_frames_gpu(self, frames: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 RGB frames [B, H, W, 3] resident on the GPU → pixel_values [B, 6, 224, 224] bf16, bit-identical to
        `apply_transform(PIL frame).to(torch.bfloat16)`: Pillow's 8-bit bicubic resize as two HBM-bound passes
        (bl_resample_pass_u8; skipped for frames already at the model resolution), then to_tensor + both normalisations +
        channel stack in one kernel (bl_preprocess_u8_bf16)."""
        from ... import ops
        (_, h, w) = self.input_sizes[0]
        if len(self.means) != 2 or any(tuple(sz) != (3, h, w) for sz in self.input_sizes):
            raise ValueError("preprocess_frames_gpu serves the fused backbone with equal input sizes")
        if frames.shape[1:3] != (h, w):
            frames = ops.resize_bicubic_u8(frames, h, w)
        key = frames.device
        if getattr(self, "_mean_std", None) is None or self._mean_std.device != key:
            flat = [v for m in self.means for v in m] + [v for s_ in self.stds for v in s_]
            self._mean_std = torch.tensor(flat, dtype=torch.float32, device=key)
        if out is None:
            out = torch.empty(frames.shape[0], 6, h, w, dtype=torch.bfloat16, device=key)
        ops.preprocess_u8(frames, self._mean_std, out)
        return out

    # ---- preprocessor_config.json (the file AutoImageProcessor reads; reference ImageProcessingMixin fields :62-126) ----
    def to_dict(self) -> Dict[str, Any]:
        return {"image_processor_type": "PrismaticImageProcessor", "processor_class": "PrismaticProcessor",
                "use_fused_vision_backbone": self.use_fused_vision_backbone, "image_resize_strategy": self.image_resize_strategy,
                "input_sizes": [list(s) for s in self.input_sizes], "means": [list(m) for m in self.means],
                "stds": [list(s) for s in self.stds]}

    def save_pretrained(self, save_directory, **_: Any) -> None:
        import json
        from pathlib import Path
        Path(save_directory).mkdir(parents=True, exist_ok=True)
        (Path(save_directory) / "preprocessor_config.json").write_text(json.dumps(self.to_dict(), indent=2))

    @classmethod
    def from_pretrained(cls, path, **_: Any) -> "PrismaticImageProcessor":
        import json
        from pathlib import Path
        f = Path(path) / "preprocessor_config.json"
        if not f.exists():
            return cls()
        raw = json.loads(f.read_text())
        sizes = raw.get("input_sizes")
        return cls(use_fused_vision_backbone=raw.get("use_fused_vision_backbone", True),
                   image_resize_strategy=raw.get("image_resize_strategy", "resize-naive"),
                   input_sizes=[tuple(s) for s in sizes] if sizes else None, means=raw.get("means"), stds=raw.get("stds"))

    def preprocess(self, images: Union[Image.Image, List[Image.Image]], return_tensors: Optional[str] = None,
                   **_: Any) -> Dict[str, Any]:
        if not isinstance(images, list):
            images = [images]
        pv = torch.stack([self.apply_transform(im.convert("RGB")) for im in images])
        return {"pixel_values": pv if return_tensors == "pt" else pv.numpy()}

    __call__ = preprocess


class ProcessorOutput(dict):
    """dict with attribute access and `.to(device, dtype=…)` (floating tensors only are cast), like HF BatchFeature."""
    __getattr__ = dict.__getitem__

    def to(self, device=None, dtype: Optional[torch.dtype] = None) -> "ProcessorOutput":
        out = ProcessorOutput()
        for k, v in self.items():
            if torch.is_tensor(v):
                v = v.to(device=device, dtype=dtype if (dtype is not None and v.is_floating_point()) else None)
            out[k] = v
        return out


class PrismaticProcessor:
    """`processor(text, images)` → input_ids, attention_mask, pixel_values (reference :187-216)."""

    def __init__(self, image_processor: Optional[PrismaticImageProcessor] = None, tokenizer: Any = None) -> None:
        self.image_processor = image_processor if image_processor is not None else PrismaticImageProcessor()
        self.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, path, **_: Any) -> "PrismaticProcessor":
        """`AutoProcessor.from_pretrained(local_dir, trust_remote_code=True)` of the reference's callers
        (run_openvla_demo.py:21, finetune.py:157): image-processor settings from preprocessor_config.json, the tokenizer
        from the directory's tokenizer files (or the synthetic stand-in when the directory was written without one)."""
        from ...util.synthetic_tokenizer import load_tokenizer
        from pathlib import Path
        has_tok = any((Path(path) / n).exists() for n in ("tokenizer.json", "tokenizer.model"))
        return cls(PrismaticImageProcessor.from_pretrained(path), load_tokenizer(path if has_tok else "synthetic"))

    def save_pretrained(self, save_directory, **_: Any) -> None:
        self.image_processor.save_pretrained(save_directory)
        if self.tokenizer is not None and hasattr(self.tokenizer, "save_pretrained"):
            self.tokenizer.save_pretrained(save_directory)

    def __call__(self, text: Union[str, List[str]], images: Union[Image.Image, List[Image.Image]],
                 return_tensors: str = "pt", **tok_kwargs: Any) -> ProcessorOutput:
        pv = self.image_processor(images, return_tensors=return_tensors)["pixel_values"]
        enc = self.tokenizer(text, return_tensors=return_tensors, **tok_kwargs)
        if pv.shape[0] != enc["input_ids"].shape[0]:
            raise ValueError("Batch is malformed; expected same number of images and text inputs!")
        return ProcessorOutput(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], pixel_values=pv)
