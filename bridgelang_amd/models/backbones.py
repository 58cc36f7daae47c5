"""The reference's backbone plugin interfaces (`VisionBackbone` prismatic/models/backbones/vision/base_vision.py:54-90,
`LLMBackbone` …/llm/base_llm.py:37-97) as thin descriptors over the HIP engine: they carry the ids, geometry, prompt
builder and transforms the training / loading scripts read, and — once bound to a VLM's weight arena — run their part of
the path (`forward`). Parameters live in the VLM's `VLAWeights`, not in the backbone objects."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Sequence, Tuple, Union

import torch

from ..extern.hf.processing_prismatic import PrismaticImageProcessor
from .prompting import PurePromptBuilder


class VisionBackbone:
    def __init__(self, vision_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224) -> None:
        self.identifier, self.image_resize_strategy, self.default_image_size = vision_backbone_id, image_resize_strategy, default_image_size
        self._vlm = None

    def get_image_transform(self) -> Callable:
        return self.image_transform

    def _bound(self):
        if self._vlm is None:
            raise RuntimeError(f"{type(self).__name__}.forward: bind the backbone to a VLM first (get_vlm / load_vla)")
        return self._vlm


class DinoSigLIPViTBackbone(VisionBackbone):
    """dinosiglip_vit.py:43-147: DINOv2-L/14-reg4 + SigLIP-SO400M/14 at 224 px, patch features concatenated."""

    def __init__(self, vision_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224) -> None:
        super().__init__(vision_backbone_id, image_resize_strategy, default_image_size)
        if default_image_size != 224:
            raise ValueError("only the 224 px fused backbone is on the OpenVLA path")
        if image_resize_strategy not in ("resize-naive", "resize-crop", "letterbox"):
            raise ValueError(f"Image Resize Strategy `{image_resize_strategy}` is not supported!")
        proc = PrismaticImageProcessor(use_fused_vision_backbone=True, image_resize_strategy=image_resize_strategy)

        def transform(img) -> Dict[str, torch.Tensor]:            # DinoSigLIPImageTransform.__call__ (:33-40)
            both = proc.apply_transform(img)
            return {"dino": both[:3], "siglip": both[3:]}
        self.image_transform = transform

    @property
    def default_image_resolution(self) -> Tuple[int, int, int]:
        return (3, self.default_image_size, self.default_image_size)

    @property
    def embed_dim(self) -> int:
        return 1024 + 1152

    @property
    def num_patches(self) -> int:
        return 256

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16

    def get_fsdp_wrapping_policy(self) -> Sequence[str]:
        """Unit boundaries (dinosiglip_vit.py:136-140: one unit per ViT block + the whole ViT): the gradient buckets."""
        return ("vision.<tower>.blockNN", "vision.<tower>.stem")

    def forward(self, pixel_values: Union[torch.Tensor, Dict[str, torch.Tensor]]) -> torch.Tensor:
        """{"dino","siglip"} [B,3,224,224] (or the stacked [B,6,224,224]) → [B, 256, 2176] bf16 patch features."""
        return self._bound().vision_features(pixel_values)

    __call__ = forward


class LLMBackbone:
    def __init__(self, llm_backbone_id: str) -> None:
        self.identifier = llm_backbone_id
        self._vlm = None


class LLaMa2LLMBackbone(LLMBackbone):
    """llama2.py:55-102 / base_llm.py:101-223 for `llama2-7b-pure` (hidden 4096, 32 layers) and `llama2-13b-pure`
    (hidden 5120, 40 layers; llama2.py:22-29): vocab padded to 32064."""
    _WIDTHS = {"llama2-7b-pure": (4096, 32), "llama2-13b-pure": (5120, 40)}

    def __init__(self, llm_backbone_id: str, llm_max_length: int = 2048, hf_token: Optional[str] = None,
                 inference_mode: bool = False, use_flash_attention_2: bool = True, tokenizer: Any = None) -> None:
        super().__init__(llm_backbone_id)
        self.llm_max_length, self.inference_mode, self.tokenizer = llm_max_length, inference_mode, tokenizer

    def get_tokenizer(self) -> Any:
        return self.tokenizer

    @property
    def prompt_builder_fn(self):
        return PurePromptBuilder                                  # llama2.py:85-93: "-pure" ids

    @property
    def transformer_layer_cls(self) -> str:
        return "LlamaDecoderLayer"

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16

    @property
    def last_layer_finetune_modules(self) -> Sequence[str]:
        n = self._WIDTHS[self.identifier][1] if self._vlm is None else self._vlm.dims.llm_layers
        return ("language_model.model.embed_tokens", f"language_model.model.layers.{n - 1}", "language_model.lm_head")

    @property
    def embed_dim(self) -> int:
        return self._WIDTHS[self.identifier][0] if self._vlm is None else self._vlm.dims.llm_dim

    @property
    def pad_token_id(self) -> int:
        return 32000

    def enable_gradient_checkpointing(self) -> None:
        """Nothing to wrap: the training strategy plans the replay itself (TrainStep(recompute=True), chosen when the
        saved activations would not fit in HBM — training/strategy.py)."""

    def get_fsdp_wrapping_policy(self) -> Sequence[str]:
        return ("llm.layerNN", "llm.lm_head")

    def embed_input_ids(self, input_ids: torch.LongTensor) -> torch.Tensor:
        if self._vlm is None:
            raise RuntimeError("bind the backbone to a VLM first")
        return self._vlm.hf.weights.embed[input_ids.to(self._vlm.device)]
