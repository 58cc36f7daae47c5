"""Batch layout of the VLA training path: `PaddedCollatorForActionPrediction` (prismatic/util/data_utils.py:94-142).

Right-pads `input_ids` with `pad_token_id` and `labels` with -100, truncates to `model_max_length`, derives
`attention_mask = input_ids != pad_token_id`, stacks `pixel_values` (tensor or dict of tensors). Host-side index work.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence

import torch

IGNORE_INDEX = -100


def _pad_right(seqs: Sequence[torch.Tensor], value: int) -> torch.Tensor:
    width = max(int(s.shape[0]) for s in seqs)
    out = torch.full((len(seqs), width), value, dtype=seqs[0].dtype)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    return out


@dataclass
class PaddedCollatorForActionPrediction:
    model_max_length: int
    pad_token_id: int
    padding_side: str = "right"
    pixel_values_dtype: torch.dtype = torch.float32

    def __call__(self, instances: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        assert self.padding_side == "right", f"Invalid Tokenizer `{self.padding_side = }`"
        input_ids = _pad_right([x["input_ids"] for x in instances], self.pad_token_id)[:, : self.model_max_length]
        labels = _pad_right([x["labels"] for x in instances], IGNORE_INDEX)[:, : self.model_max_length]
        pixels = [x["pixel_values"] for x in instances]
        assert all(p is not None for p in pixels), "Invalid VLA Example with `pixel_values = None`!"
        if isinstance(pixels[0], torch.Tensor):
            pixel_values = torch.stack(pixels)
        elif isinstance(pixels[0], dict):
            pixel_values = {k: torch.stack([p[k] for p in pixels]) for k in pixels[0]}
        else:
            raise ValueError(f"Unsupported `pixel_values` type = {type(pixels)}")
        out = dict(pixel_values=pixel_values, input_ids=input_ids, attention_mask=input_ids.ne(self.pad_token_id),
                   labels=labels)
        if "dataset_name" in instances[0]:
            out["dataset_names"] = [x["dataset_name"] for x in instances]
        return out
