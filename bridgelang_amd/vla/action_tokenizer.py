"""ActionTokenizer — the 256-bin action ↔ token-id map of the reference (prismatic/vla/action_tokenizer.py:13-72).

Host-side integer/f64 work (7 values per sample); kept in numpy. Same constructor, `__call__`,
`decode_token_ids_to_actions`, `vocab_size`, `bins`, `bin_centers`, `action_token_begin_idx` as the reference. The
tokenizer only needs `.vocab_size`, `.decode(list[int])` and `.batch_decode(list[list[int]])`.
"""
from __future__ import annotations

from typing import List, Union

import numpy as np


class ActionTokenizer:
    def __init__(self, tokenizer, bins: int = 256, min_action: int = -1, max_action: int = 1) -> None:
        self.tokenizer, self.n_bins, self.min_action, self.max_action = tokenizer, bins, min_action, max_action
        self.bins = np.linspace(min_action, max_action, self.n_bins)          # 256 edges → 255 intervals
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0
        # the last n_bins ids of the vocabulary are action tokens; everything above this index is an action token
        self.action_token_begin_idx: int = int(self.tokenizer.vocab_size - (self.n_bins + 1))

    def encode_ids(self, action: np.ndarray) -> np.ndarray:
        """ids = vocab_size - digitize(clip(action)) ∈ [vocab-256, vocab-1] (no string round trip)."""
        clipped = np.clip(action, a_min=float(self.min_action), a_max=float(self.max_action))
        return self.tokenizer.vocab_size - np.digitize(clipped, self.bins)

    def __call__(self, action: np.ndarray) -> Union[str, List[str]]:
        ids = self.encode_ids(action)
        if ids.ndim == 1:
            return self.tokenizer.decode(list(ids))
        return self.tokenizer.batch_decode(ids.tolist())

    def decode_token_ids_to_actions(self, action_token_ids: np.ndarray) -> np.ndarray:
        # digitize yields 1..256 but only 255 intervals exist: clip index 255 onto the last centre (reference :49-68)
        idx = np.clip(self.tokenizer.vocab_size - action_token_ids - 1, a_min=0, a_max=self.bin_centers.shape[0] - 1)
        return self.bin_centers[idx]

    @property
    def vocab_size(self) -> int:
        return self.n_bins
