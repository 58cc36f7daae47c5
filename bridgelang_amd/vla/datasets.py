"""Training-sample construction for the VLA path: `RLDSBatchTransform` and `DummyDataset`
(prismatic/vla/datasets/datasets.py:30-67,180-232). A sample is

    prompt  = "In: What action should the robot take to {instruction}?\\nOut: {7 action tokens}</s>"
    labels  = input_ids with everything but the last (action_dim + 1) positions set to -100   (:63, :230)

The RLDS/TFDS reader that feeds `RLDSBatchTransform` in the reference is out of scope; the transform
itself only needs the dict it documents (`dataset_name`, `action[0]`, `observation.image_primary[0]`,
`task.language_instruction`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch
from PIL import Image

from ..models.prompting import PurePromptBuilder
from .action_tokenizer import ActionTokenizer

IGNORE_INDEX = -100


def build_sample(action_tokenizer: ActionTokenizer, base_tokenizer: Any, image_transform: Callable[[Image.Image], Any],
                 prompt_builder_fn: Callable[[str], Any], image: Image.Image, instruction: str, action: np.ndarray,
                 predict_stop_token: bool = True) -> Dict[str, Any]:
    pb = prompt_builder_fn("openvla")
    pb.add_turn("human", f"What action should the robot take to {instruction}?")
    pb.add_turn("gpt", action_tokenizer(action))
    ids = torch.tensor(base_tokenizer(pb.get_prompt(), add_special_tokens=True).input_ids)
    labels = ids.clone()
    labels[: -(len(action) + 1)] = IGNORE_INDEX          # loss only on the action tokens (+ stop token)
    if not predict_stop_token:
        labels[-1] = IGNORE_INDEX
    return dict(pixel_values=image_transform(image), input_ids=ids, labels=labels)


@dataclass
class RLDSBatchTransform:
    action_tokenizer: ActionTokenizer
    base_tokenizer: Any
    image_transform: Callable[[Image.Image], Any]
    prompt_builder_fn: Callable[[str], Any] = PurePromptBuilder
    predict_stop_token: bool = True

    def __call__(self, rlds_batch: Dict[str, Any]) -> Dict[str, Any]:
        img = Image.fromarray(rlds_batch["observation"]["image_primary"][0])
        lang = rlds_batch["task"]["language_instruction"].decode().lower()
        out = build_sample(self.action_tokenizer, self.base_tokenizer, self.image_transform, self.prompt_builder_fn, img,
                           lang, rlds_batch["action"][0], self.predict_stop_token)
        out["dataset_name"] = rlds_batch["dataset_name"]
        return out


class DummyDataset(torch.utils.data.Dataset):
    """Synthetic 224×224 frames + uniform[0,1)^7 actions + a fixed instruction; identity q01/q99 statistics."""

    def __init__(self, action_tokenizer: ActionTokenizer, base_tokenizer: Any,
                 image_transform: Callable[[Image.Image], Any], prompt_builder_fn: Callable[[str], Any] = PurePromptBuilder,
                 length: int = 10000, seed: Optional[int] = None) -> None:
        self.action_tokenizer, self.base_tokenizer = action_tokenizer, base_tokenizer
        self.image_transform, self.prompt_builder_fn = image_transform, prompt_builder_fn
        self.length, self.seed = length, seed
        self.dataset_statistics = {"dummy_dataset": {"action": {"q01": np.zeros((7,), dtype=np.float32),
                                                                "q99": np.ones((7,), dtype=np.float32)}}}

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        rng = np.random if self.seed is None else np.random.RandomState(self.seed + idx)   # seedable (reference: global RNG)
        image = Image.fromarray(np.asarray(rng.rand(224, 224, 3) * 255.0, dtype=np.uint8))
        action = np.asarray(rng.rand(7), dtype=np.float32)
        return build_sample(self.action_tokenizer, self.base_tokenizer, self.image_transform, self.prompt_builder_fn,
                            image, "do something spectacular", action)


class EpochIterable(torch.utils.data.IterableDataset):
    """A map-style dataset presented as the endless IterableDataset `run_vla_training` expects (the RLDS loader repeats
    forever, base_strategy.py:259-266); `len()` is one epoch, which sizes the LR schedule."""

    def __init__(self, dataset) -> None:
        self.dataset = dataset
        self.dataset_statistics = getattr(dataset, "dataset_statistics", None)

    def __len__(self) -> int:
        return len(self.dataset)

    def __iter__(self):
        while True:
            for i in range(len(self.dataset)):
                yield self.dataset[i]
