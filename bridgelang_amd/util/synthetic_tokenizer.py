"""A deterministic stand-in for the Llama-2 tokenizer for SYNTHETIC runs (benchmarks, offline smoke runs of the entry
scripts): no tokenizer files exist offline and nothing may be fetched, and the hot path only needs ids with Llama's
special-token layout (BOS 1, EOS 2, vocabulary 32000, pad id 32000 appended by the reference at llama2.py:74-76).
One id per whitespace-separated piece (a stable hash into [3, 31743), below the 256 action-token ids); integer pieces
map to themselves, which is what `decode()` emits, so `ActionTokenizer.__call__` → text → ids round-trips exactly as it
does through the real tokenizer (action_tokenizer.py:38-47). Not a language tokenizer: real checkpoints ship their own
(`AutoTokenizer.from_pretrained(local_dir)`), which the scripts use whenever the files are present."""
from __future__ import annotations

import zlib
from types import SimpleNamespace
from typing import Any, List, Sequence, Union

import torch


class SyntheticLlamaTokenizer:
    vocab_size, bos_token_id, eos_token_id, pad_token_id = 32000, 1, 2, 32000
    model_max_length, padding_side = 2048, "right"

    def _piece(self, w: str) -> int:
        if w == "</s>":
            return self.eos_token_id
        if w.isdigit() and int(w) < self.vocab_size:
            return int(w)
        return 3 + zlib.crc32(w.encode()) % (31743 - 3)

    def _ids(self, text: str, add_special_tokens: bool = True) -> List[int]:
        body = [self._piece(w) for w in text.replace("</s>", " </s> ").split()]
        return ([self.bos_token_id] if add_special_tokens else []) + body

    def decode(self, ids: Sequence[int], **_: Any) -> str:
        return " ".join(str(int(i)) for i in ids)

    def batch_decode(self, rows: Sequence[Sequence[int]], **_: Any) -> List[str]:
        return [self.decode(r) for r in rows]

    def __call__(self, text: Union[str, List[str]], add_special_tokens: bool = True, return_tensors: Any = None,
                 truncation: bool = False, **_: Any):
        if isinstance(text, str) and return_tensors is None:
            return SimpleNamespace(input_ids=self._ids(text, add_special_tokens))
        rows = [self._ids(t, add_special_tokens) for t in ([text] if isinstance(text, str) else text)]
        n = max(len(r) for r in rows)
        ids = torch.tensor([r + [self.pad_token_id] * (n - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": (ids != self.pad_token_id).long()}

    def save_pretrained(self, save_directory, **_: Any) -> None:
        from pathlib import Path
        Path(save_directory).mkdir(parents=True, exist_ok=True)
        (Path(save_directory) / "synthetic_tokenizer.json").write_text('{"tokenizer": "SyntheticLlamaTokenizer"}')


def load_tokenizer(path_or_flag: Any = None):
    """`synthetic` (or a directory written by SyntheticLlamaTokenizer.save_pretrained) → the stand-in; a directory with
    real tokenizer files → `AutoTokenizer.from_pretrained(dir)` + the reference's pad token (llama2.py:74-76)."""
    from pathlib import Path
    if path_or_flag in (None, "synthetic") or (Path(str(path_or_flag)) / "synthetic_tokenizer.json").exists():
        return SyntheticLlamaTokenizer()
    d = Path(str(path_or_flag))
    if not d.is_dir():
        raise FileNotFoundError(f"tokenizer `{path_or_flag}`: not a local directory (nothing is fetched from the hub); "
                                f"pass `synthetic` for synthetic runs")
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(str(d), model_max_length=2048, padding_side="right")
    if tok.pad_token_id is None:
        tok.add_special_tokens({"pad_token": "<PAD>"})
    return tok
