"""Backbone / VLM registries under the reference's ids (prismatic/models/materialize.py:29-130): an MI355X backbone
registers as `dinosiglip-vit-so-224px` / `llama2-7b-pure`, so `get_vision_backbone_and_transform`,
`get_llm_backbone_and_tokenizer` and `get_vlm` keep their call shapes (vla-scripts/train.py:150-180, load.py:196-208).
Ids outside the OpenVLA path raise the reference's ValueError."""
from __future__ import annotations

from typing import Any, Callable, Optional, Tuple

from .backbones import DinoSigLIPViTBackbone, LLaMa2LLMBackbone, LLMBackbone, VisionBackbone
from .vlms import PrismaticVLM

VISION_BACKBONES = {
    "dinosiglip-vit-so-224px": {"cls": DinoSigLIPViTBackbone, "kwargs": {"default_image_size": 224}},
}
LLM_BACKBONES = {
    "llama2-7b-pure": {"cls": LLaMa2LLMBackbone, "kwargs": {}},
    "llama2-13b-pure": {"cls": LLaMa2LLMBackbone, "kwargs": {}},       # BASELINE configs[4] (models/materialize.py:57)
}


def get_vision_backbone_and_transform(vision_backbone_id: str, image_resize_strategy: str) -> Tuple[VisionBackbone, Callable]:
    if vision_backbone_id in VISION_BACKBONES:
        cfg = VISION_BACKBONES[vision_backbone_id]
        vision_backbone = cfg["cls"](vision_backbone_id, image_resize_strategy, **cfg["kwargs"])
        return vision_backbone, vision_backbone.get_image_transform()
    raise ValueError(f"Vision Backbone `{vision_backbone_id}` is not supported!")


def get_llm_backbone_and_tokenizer(llm_backbone_id: str, llm_max_length: int = 2048, hf_token: Optional[str] = None,
                                   inference_mode: bool = False, tokenizer: Any = None) -> Tuple[LLMBackbone, Any]:
    """`tokenizer`: the Llama tokenizer object (the reference downloads it, base_llm.py:139-151; there is no network
    here, so the caller supplies it — None is allowed for id-level pipelines)."""
    if llm_backbone_id in LLM_BACKBONES:
        cfg = LLM_BACKBONES[llm_backbone_id]
        llm_backbone = cfg["cls"](llm_backbone_id, llm_max_length=llm_max_length, hf_token=hf_token,
                                  inference_mode=inference_mode, tokenizer=tokenizer, **cfg["kwargs"])
        return llm_backbone, llm_backbone.get_tokenizer()
    raise ValueError(f"LLM Backbone `{llm_backbone_id}` is not supported!")


def get_vlm(model_id: str, arch_specifier: str, vision_backbone: VisionBackbone, llm_backbone: LLMBackbone,
            enable_mixed_precision_training: bool = True, **kwargs: Any) -> PrismaticVLM:
    return PrismaticVLM(model_id, vision_backbone, llm_backbone, enable_mixed_precision_training=enable_mixed_precision_training,
                        arch_specifier=arch_specifier, **kwargs)
