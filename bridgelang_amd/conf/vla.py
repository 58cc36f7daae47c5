"""`VLAConfig` and its registry of named VLA training configurations (prismatic/conf/vla.py:20-57,202-235): same field
names, defaults and `vla_id`s, so `--vla.type prism-dinosiglip-224px+mx-bridge` and `--vla.<field> <value>` mean what
they mean to the reference's `vla-scripts/train.py`. Only configurations whose base VLM is on the MI355X path (fused
DINOv2 + SigLIP at 224 px with Llama-2) are registered; the SigLIP-only / ablation ids of the reference raise the same
"unknown choice" error any unregistered id raises. Two additions, marked: the FSDP shard-grad-op twin of the Bridge
config (BASELINE configs[2]) and the 13B composition (BASELINE configs[4])."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, unique
from pathlib import Path
from typing import Dict, Optional, Type, Union


@dataclass
class VLAConfig:
    # fmt: off
    vla_id: str                                     # unique id of the configuration variant
    base_vlm: Union[str, Path]                      # base VLM id or path to a run directory
    freeze_vision_backbone: bool
    freeze_llm_backbone: bool
    unfreeze_last_llm_layer: bool
    data_mix: str                                   # Open-X mixture id (`bridge`, `oxe_magic_soup_plus_minus`, …)
    shuffle_buffer_size: int
    epochs: int
    max_steps: Optional[int]
    expected_world_size: int
    global_batch_size: int
    per_device_batch_size: int
    learning_rate: float
    weight_decay: float
    max_grad_norm: float
    lr_scheduler_type: str
    warmup_ratio: float
    train_strategy: str
    enable_gradient_checkpointing: bool = True
    enable_mixed_precision_training: bool = True
    reduce_in_full_precision: bool = True
    # fmt: on


    @classmethod
    def register_subclass(cls, name: str, sub: "Type[VLAConfig]") -> None:
        _REGISTRY[name] = sub

    @classmethod
    def get_choice_class(cls, name: str) -> "Type[VLAConfig]":
        if name not in _REGISTRY:
            raise KeyError(f"Couldn't find a choice class for '{name}' in {sorted(_REGISTRY)}")
        return _REGISTRY[name]

    @classmethod
    def get_known_choices(cls) -> "Dict[str, Type[VLAConfig]]":
        return dict(_REGISTRY)


_REGISTRY: Dict[str, Type[VLAConfig]] = {}


# [8 GPU] DINO-SigLIP 224px + Bridge (conf/vla.py:103-108 over :62-91)
@dataclass
class Exp_DinoSigLIP_224px_Bridge(VLAConfig):
    vla_id: str = "prism-dinosiglip-224px+mx-bridge"
    base_vlm: Union[str, Path] = "prism-dinosiglip-224px+7b"
    freeze_vision_backbone: bool = False
    freeze_llm_backbone: bool = False
    unfreeze_last_llm_layer: bool = False
    data_mix: str = "bridge"
    shuffle_buffer_size: int = 256_000
    epochs: int = 1000
    max_steps: Optional[int] = None
    expected_world_size: int = 8
    global_batch_size: int = 256
    per_device_batch_size: int = 32
    learning_rate: float = 2e-5
    weight_decay: float = 0.0
    max_grad_norm: float = 1.0
    lr_scheduler_type: str = "constant"
    warmup_ratio: float = 0.0
    train_strategy: str = "fsdp-full-shard"


# [64 GPU] OpenVLA 7B: DINO-SigLIP 224px + OXE Magic Soup++ (conf/vla.py:128-141)
@dataclass
class Exp_DinoSigLIP_224px_OXE_Magic_Soup_Plus(Exp_DinoSigLIP_224px_Bridge):
    vla_id: str = "prism-dinosiglip-224px+mx-oxe-magic-soup-plus"
    data_mix: str = "oxe_magic_soup_plus_minus"
    shuffle_buffer_size: int = 1_000_000
    expected_world_size: int = 64
    global_batch_size: int = 2048
    per_device_batch_size: int = 32


# (addition) BASELINE configs[2]: the Bridge configuration under FSDP shard-grad-op on one 8-GPU node
@dataclass
class Exp_DinoSigLIP_224px_Bridge_ShardGradOp(Exp_DinoSigLIP_224px_Bridge):
    vla_id: str = "prism-dinosiglip-224px+mx-bridge+shard-grad-op"
    train_strategy: str = "fsdp-shard-grad-op"


# (addition) BASELINE configs[4]: dinosiglip-vit-so-224px + llama2-13b-pure, FSDP full-shard (not a registered reference
# config — SURVEY §8d cfg 5 — composed from the reference's backbone registries, models/materialize.py:35,57)
@dataclass
class Exp_DinoSigLIP_224px_13B_Bridge(Exp_DinoSigLIP_224px_Bridge):
    vla_id: str = "prism-dinosiglip-224px-13b+mx-bridge"
    base_vlm: Union[str, Path] = "prism-dinosiglip-224px+13b"
    per_device_batch_size: int = 16
    global_batch_size: int = 128


@unique
class VLARegistry(Enum):
    DINOSIGLIP_224PX_MX_BRIDGE = Exp_DinoSigLIP_224px_Bridge
    DINOSIGLIP_224PX_MX_OXE_MAGIC_SOUP_PLUS = Exp_DinoSigLIP_224px_OXE_Magic_Soup_Plus
    DINOSIGLIP_224PX_MX_BRIDGE_SHARD_GRAD_OP = Exp_DinoSigLIP_224px_Bridge_ShardGradOp
    DINOSIGLIP_224PX_13B_MX_BRIDGE = Exp_DinoSigLIP_224px_13B_Bridge

    @property
    def vla_id(self) -> str:
        return self.value.vla_id


for _variant in VLARegistry:
    VLAConfig.register_subclass(_variant.vla_id, _variant.value)
