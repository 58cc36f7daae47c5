"""`PrismaticConfig` / `OpenVLAConfig` for the MI355X path.

Same constructor keywords, attribute names and defaults as the reference's HF config
(prismatic/extern/hf/configuration_prismatic.py:72-140) so `config.json` files written by either side load on the
other: `vision_backbone_id`, `llm_backbone_id`, `arch_specifier`, `use_fused_vision_backbone`, `image_resize_strategy`,
`text_config`, `llm_max_length`, `pad_token_id`, `pad_to_multiple_of`, `output_projector_states`, `norm_stats`,
`n_action_bins`, plus the derived `timm_model_ids`, `timm_override_act_layers`, `image_sizes`, `hf_llm_id`.
Only the backbones on the OpenVLA hot path are registered here (fused DINOv2+SigLIP at 224 px, Llama-2 7B/13B);
asking for another id raises the same ValueError the reference raises for an unknown id.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

from transformers import PretrainedConfig
from transformers.models.auto import CONFIG_MAPPING

# backbone id → (image sizes, timm ids, act-layer overrides); reference tables :15-45
_VISION = {
    "dinosiglip-vit-so-224px": ([224, 224], ["vit_large_patch14_reg4_dinov2.lvd142m", "vit_so400m_patch14_siglip_224"],
                                [None, None]),
}
# llm id → (HF hub path, HF model family); reference tables :47-65
_LLM = {
    "llama2-7b-pure": ("meta-llama/Llama-2-7b-hf", "llama"),
    "llama2-13b-pure": ("meta-llama/Llama-2-13b-hf", "llama"),
}
VALID_VISION_BACKBONES = set(_VISION)
VALID_LLM_BACKBONES = set(_LLM)


class PrismaticConfig(PretrainedConfig):
    model_type: str = "prismatic"
    is_composition: bool = False

    def __init__(self, vision_backbone_id: str = "dinosiglip-vit-so-224px", llm_backbone_id: str = "llama2-7b-pure",
                 arch_specifier: str = "no-align+fused-gelu-mlp", use_fused_vision_backbone: Optional[bool] = None,
                 image_resize_strategy: str = "resize-naive", text_config: Optional[Dict[str, Any]] = None,
                 llm_max_length: int = 2048, pad_token_id: int = 32000, pad_to_multiple_of: int = 64,
                 output_projector_states: bool = False, **kwargs: Any) -> None:
        if vision_backbone_id not in VALID_VISION_BACKBONES:
            raise ValueError(f"Vision backbone `{vision_backbone_id}` not in {VALID_VISION_BACKBONES = }")
        if llm_backbone_id not in VALID_LLM_BACKBONES:
            raise ValueError(f"LLM backbone `{llm_backbone_id}` not in {VALID_LLM_BACKBONES = }")
        self.vision_backbone_id, self.llm_backbone_id = vision_backbone_id, llm_backbone_id
        self.arch_specifier, self.output_projector_states = arch_specifier, output_projector_states
        self.use_fused_vision_backbone = (vision_backbone_id.startswith(("dinoclip", "dinosiglip"))
                                          if use_fused_vision_backbone is None else use_fused_vision_backbone)
        self.image_sizes, self.timm_model_ids, self.timm_override_act_layers = _VISION[vision_backbone_id]
        self.image_resize_strategy = image_resize_strategy
        self.hf_llm_id, family = _LLM[llm_backbone_id]
        self.llm_max_length = llm_max_length
        self.pad_token_id, self.pad_to_multiple_of = pad_token_id, pad_to_multiple_of
        llm_cls = CONFIG_MAPPING[family]
        if text_config is None:
            # the reference's conversion script starts from a default LlamaConfig and grows the vocabulary to 32064
            # (convert_openvla_weights_to_hf.py:152-160,174-176); 13B needs its own widths
            extra = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                         num_key_value_heads=40) if "13b" in llm_backbone_id else {}
            text_config = dict(vocab_size=32000 + pad_to_multiple_of, pad_token_id=pad_token_id, rms_norm_eps=1e-6,
                               **extra)
        self.text_config = llm_cls(**text_config) if isinstance(text_config, dict) else text_config
        super().__init__(pad_token_id=pad_token_id, **kwargs)


class OpenVLAConfig(PrismaticConfig):
    model_type: str = "openvla"

    def __init__(self, norm_stats: Optional[Dict[str, Dict[str, Dict[str, Dict[str, List[float]]]]]] = None,
                 n_action_bins: int = 256, **kwargs: Any) -> None:
        self.norm_stats, self.n_action_bins = norm_stats, n_action_bins
        super().__init__(**kwargs)
