"""Checkpoint naming between the HF state dict this package works in and the reference's native training checkpoints
(`{"model": {"vision_backbone": …, "projector": …, "llm_backbone": …}}`, fsdp.py:95-133), i.e. the inverse of
vla-scripts/extern/convert_openvla_weights_to_hf.py:73-115 so files written here go through the reference's own
conversion / loading code unchanged."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

_PROJ = {"projector.fc1.": "projector.0.", "projector.fc2.": "projector.2.", "projector.fc3.": "projector.4."}


def hf_to_prismatic(name: str) -> Tuple[str, str]:
    """HF name → (module key, key inside that module's state dict)."""
    if name.startswith("projector."):
        for a, b in _PROJ.items():
            if name.startswith(a):
                return "projector", b + name[len(a):]
    if name.startswith("language_model."):
        return "llm_backbone", "llm." + name[len("language_model."):]
    if name.startswith("vision_backbone.featurizer."):
        key = "dino_featurizer." + name[len("vision_backbone.featurizer."):]
        if key.endswith(".scale_factor"):
            key = key[:-len(".scale_factor")] + ".gamma"          # timm LayerScale name (HF file patches it)
        return "vision_backbone", key
    if name.startswith("vision_backbone.fused_featurizer."):
        return "vision_backbone", "siglip_featurizer." + name[len("vision_backbone.fused_featurizer."):]
    raise KeyError(name)


def prismatic_to_hf(module: str, key: str) -> str:
    if module == "projector":
        for a, b in _PROJ.items():
            if key.startswith(b):
                return a + key[len(b):]
    elif module == "llm_backbone" and key.startswith("llm."):
        return "language_model." + key[len("llm."):]
    elif module == "vision_backbone":
        if key.startswith("dino_featurizer."):
            key = key[len("dino_featurizer."):]
            if key.endswith(".gamma"):
                key = key[:-len(".gamma")] + ".scale_factor"
            return "vision_backbone.featurizer." + key
        if key.startswith("siglip_featurizer."):
            return "vision_backbone.fused_featurizer." + key[len("siglip_featurizer."):]
    raise KeyError((module, key))


def to_model_state_dicts(hf_sd: Dict[str, torch.Tensor], module_keys) -> Dict[str, Dict[str, torch.Tensor]]:
    out: Dict[str, Dict[str, torch.Tensor]] = {m: {} for m in module_keys}
    for name, t in hf_sd.items():
        m, k = hf_to_prismatic(name)
        if m in out:
            out[m][k] = t
    return out


def from_model_state_dicts(model: Dict[str, Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    return {prismatic_to_hf(m, k): t for m, sd in model.items() for k, t in sd.items()}
