"""Native model classes of the reference — `PrismaticVLM` (prismatic/models/vlms/prismatic.py:36-518) and `OpenVLA`
(prismatic/models/vlas/openvla.py:20-131) — over the HIP path: same constructor / forward / freeze_backbones /
predict_action(image, instruction) surface, the arithmetic delegated to the HF-interface model
(extern/hf/modeling_prismatic.py), which owns the weight arena and the engines."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from ..extern.hf.configuration_prismatic import OpenVLAConfig
from ..extern.hf.modeling_prismatic import OpenVLAForActionPrediction
from ..training.checkpoint import from_model_state_dicts, to_model_state_dicts
from ..training.step import STAGES
from ..weights import VLADims


class PrismaticVLM:
    def __init__(self, model_id: str, vision_backbone, llm_backbone, enable_mixed_precision_training: bool = True,
                 arch_specifier: str = "no-align+fused-gelu-mlp", device: Union[str, torch.device] = "cuda:0",
                 dims: Optional[VLADims] = None, norm_stats: Optional[Dict[str, Any]] = None, **_: Any) -> None:
        if arch_specifier not in ("no-align+fused-gelu-mlp", "fused-gelu-mlp"):
            raise ValueError(f"PrismaticVLM with `{arch_specifier = }` is not supported!")
        self.model_id, self.arch_specifier = model_id, arch_specifier
        self.vision_backbone, self.llm_backbone = vision_backbone, llm_backbone
        self.enable_mixed_precision_training = enable_mixed_precision_training
        cfg = OpenVLAConfig(vision_backbone_id=vision_backbone.identifier, llm_backbone_id=llm_backbone.identifier,
                            arch_specifier=arch_specifier, image_resize_strategy=vision_backbone.image_resize_strategy,
                            llm_max_length=llm_backbone.llm_max_length, norm_stats=norm_stats or {})
        self.hf = OpenVLAForActionPrediction(cfg, device=device, dims=dims)
        self.device, self.dims = self.hf.device, self.hf.dims
        self.weights = self.hf.weights                       # what the training strategy reads (strategy.vlm.weights)
        vision_backbone._vlm = llm_backbone._vlm = self
        self.all_module_keys = ["vision_backbone", "llm_backbone", "projector"]
        self.trainable_module_keys: List[str] = []
        self.vision_backbone_requires_grad = False
        self.stage: Optional[str] = None

    # ---- weights ----
    @classmethod
    def from_pretrained(cls, pretrained_checkpoint: Union[str, Path], model_id: str, vision_backbone, llm_backbone,
                        enable_mixed_precision_training: bool = True, arch_specifier: str = "no-align+fused-gelu-mlp",
                        freeze_weights: bool = True, **kwargs: Any) -> "PrismaticVLM":
        """prismatic.py:96-127: `{"model": {"projector", "llm_backbone", ["vision_backbone"]}}` checkpoint."""
        vlm = cls(model_id, vision_backbone, llm_backbone, enable_mixed_precision_training=enable_mixed_precision_training,
                  arch_specifier=arch_specifier, **kwargs)
        model_state_dict = torch.load(pretrained_checkpoint, map_location="cpu", weights_only=True)["model"]
        assert "projector" in model_state_dict and "llm_backbone" in model_state_dict, \
            "PrismaticVLM `from_pretrained` expects checkpoint with keys for `projector` AND `llm_backbone`!"
        vlm.load_model_state_dicts(model_state_dict)
        return vlm

    def load_model_state_dicts(self, model_state_dict: Dict[str, Dict[str, torch.Tensor]]) -> None:
        self.hf.load_state_dict(from_model_state_dicts(model_state_dict), strict=False)

    def model_state_dicts(self, module_keys: Optional[Sequence[str]] = None) -> Dict[str, Dict[str, torch.Tensor]]:
        return to_model_state_dicts(self.hf.state_dict(), module_keys or self.all_module_keys)

    # ---- training-script surface ----
    def freeze_backbones(self, stage: str) -> None:
        """prismatic.py:129-241: records which modules train; the optimizer state is built from it by the strategy."""
        if stage not in STAGES or stage == "lora":
            raise ValueError(f"Stage `{stage}` is not supported for LLaVa! Try < align | finetune >")
        vision, proj, llm = STAGES[stage]
        self.stage = stage
        self.vision_backbone_requires_grad = vision
        self.trainable_module_keys = [k for k, on in (("vision_backbone", vision), ("projector", proj), ("llm_backbone", llm != "none")) if on]

    def get_fsdp_wrapping_policy(self) -> Sequence[str]:
        """Unit boundaries = gradient buckets of training/sharding.py (prismatic.py:285-306)."""
        return tuple(self.vision_backbone.get_fsdp_wrapping_policy()) + ("projector",) + tuple(self.llm_backbone.get_fsdp_wrapping_policy())

    def get_prompt_builder(self, system_prompt: Optional[str] = None):
        return self.llm_backbone.prompt_builder_fn(self.model_family, system_prompt=system_prompt)

    model_family = "prismatic"

    # ---- compute ----
    def _stack(self, pixel_values: Union[torch.Tensor, Dict[str, torch.Tensor]]) -> torch.Tensor:
        if isinstance(pixel_values, dict):
            pixel_values = torch.cat([pixel_values["dino"], pixel_values["siglip"]], dim=1)
        return pixel_values

    def vision_features(self, pixel_values: Union[torch.Tensor, Dict[str, torch.Tensor]]) -> torch.Tensor:
        pv = self._stack(pixel_values).to(self.device).to(torch.bfloat16)
        eng = self.hf.engine(pv.shape[0], 1)
        eng.pixel_values.copy_(pv)
        eng.run_vision()
        return eng.feats.view(pv.shape[0], self.dims.n_patches, self.dims.vision_dim).clone()

    def forward(self, input_ids: Optional[torch.LongTensor] = None, attention_mask: Optional[torch.Tensor] = None,
                pixel_values: Optional[Union[torch.Tensor, Dict[str, torch.Tensor]]] = None,
                labels: Optional[torch.LongTensor] = None, inputs_embeds: Optional[torch.FloatTensor] = None,
                past_key_values: Optional[Any] = None, use_cache: Optional[bool] = None,
                output_attentions: Optional[bool] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None, multimodal_indices: Optional[torch.LongTensor] = None):
        """prismatic.py:312-481. `multimodal_indices` must select the whole batch (the VLA path is always multimodal)."""
        if multimodal_indices is not None and len(multimodal_indices) != input_ids.shape[0]:
            raise NotImplementedError("mixed unimodal / multimodal batches are outside the VLA path")
        return self.hf.forward(input_ids=input_ids, attention_mask=attention_mask, pixel_values=self._stack(pixel_values),
                               labels=labels, inputs_embeds=inputs_embeds, past_key_values=past_key_values, use_cache=use_cache,
                               output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                               return_dict=return_dict)

    __call__ = forward

    def generate(self, input_ids: torch.LongTensor, pixel_values, max_new_tokens: int = 7, **kwargs: Any) -> torch.LongTensor:
        return self.hf.generate(input_ids, max_new_tokens=max_new_tokens, pixel_values=self._stack(pixel_values), **kwargs)


class OpenVLA(PrismaticVLM):
    def __init__(self, *args, norm_stats: Dict[str, Dict[str, Dict[str, Dict[str, List[float]]]]], action_tokenizer, **kwargs) -> None:
        super().__init__(*args, norm_stats=norm_stats, **kwargs)
        self.norm_stats, self.action_tokenizer = norm_stats, action_tokenizer

    model_family = "openvla"

    def predict_action(self, image, instruction: str, unnorm_key: Optional[str] = None, **kwargs: str) -> np.ndarray:
        """openvla.py:35-103: PIL image + instruction → un-normalised continuous action."""
        image_transform, tokenizer = self.vision_backbone.image_transform, self.llm_backbone.tokenizer
        prompt_builder = self.get_prompt_builder()
        prompt_builder.add_turn(role="human", message=f"What action should the robot take to {instruction.lower()}?")
        prompt_text = prompt_builder.get_prompt()
        input_ids = tokenizer(prompt_text, truncation=True, return_tensors="pt")["input_ids"].to(self.device)
        if not torch.all(input_ids[:, -1] == 29871):              # the special empty token after "Out:" (openvla.py:58-66)
            input_ids = torch.cat((input_ids, torch.tensor([[29871]], dtype=torch.long, device=input_ids.device)), dim=1)
        pixel_values = image_transform(image)
        if isinstance(pixel_values, torch.Tensor):
            pixel_values = pixel_values[None, ...].to(self.device)
        elif isinstance(pixel_values, dict):
            pixel_values = {k: v[None, ...].to(self.device) for k, v in pixel_values.items()}
        else:
            raise ValueError(f"Unsupported `pixel_values` type = {type(pixel_values)}")
        n = self.get_action_dim(unnorm_key)
        generated_ids = self.generate(input_ids, pixel_values, max_new_tokens=n, **kwargs)
        predicted_action_token_ids = generated_ids[0, -n:]
        normalized_actions = self.action_tokenizer.decode_token_ids_to_actions(predicted_action_token_ids.cpu().numpy())
        action_norm_stats = self.get_action_stats(unnorm_key)
        mask = action_norm_stats.get("mask", np.ones_like(action_norm_stats["q01"], dtype=bool))
        action_high, action_low = np.array(action_norm_stats["q99"]), np.array(action_norm_stats["q01"])
        return np.where(mask, 0.5 * (normalized_actions + 1) * (action_high - action_low) + action_low, normalized_actions)

    @staticmethod
    def _check_unnorm_key(norm_stats: Dict, unnorm_key: str) -> str:
        return OpenVLAForActionPrediction._check_unnorm_key(norm_stats, unnorm_key)

    def get_action_dim(self, unnorm_key: Optional[str] = None) -> int:
        return len(self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]["q01"])

    def get_action_stats(self, unnorm_key: Optional[str] = None) -> Dict:
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]
