"""The three branches of `PrismaticForConditionalGeneration.forward` (modeling_prismatic.py:322-415) over the HIP engine.

  * multimodal / language-only forward for ALL positions (+ shifted cross-entropy): the attention mask and the labels get
    256 columns inserted after column 0 (True / -100, :387-401), HF's CausalLM loss shifts labels by one and averages over
    the non-ignored tokens; optional per-layer hidden states (HF order: embeddings, every layer's
    output, the last one after the final norm);
  * generation prefill (`use_cache=True`): the generation plan of engine.py — the same launches `generate()` replays —
    returning the last position's logits and an `EngineKVCache` handle;
  * cached generation step (`input_ids [B, 1]` + `past_key_values`, :325-341): one decode step on the handle's engine.

Host code here only builds index tensors and picks plans; the arithmetic is the HIP plan in engine.py plus
bl_cross_entropy_f32.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import ops
from .engine import OpenVLAEngine

IGNORE_INDEX = -100


class EngineKVCache:
    """`past_key_values` of this path: an opaque handle on the KV caches of one engine slot (the caches are
    preallocated `[layer][B, H, cache_len, 128]` device buffers of an OpenVLAEngine, not per-call tensors). It knows how
    many tokens it holds and refuses to be used after the engine's caches were overwritten by another prefill.
    `get_seq_length()` / `len()` follow the HF Cache protocol far enough for step-wise callers."""

    def __init__(self, engine: OpenVLAEngine):
        self.engine, self.epoch = engine, engine.epoch
        self.steps = 0                       # cached decode steps run so far (tokens held = engine.S + steps)

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.engine.S + self.steps

    def __len__(self) -> int:
        return self.engine.dims.llm_layers

    def check(self) -> None:
        if self.epoch != self.engine.epoch:
            raise RuntimeError("stale past_key_values: the engine slot holding this KV cache has since run another prefill of "
                               "the same (batch, prompt length); finish one generation before starting the next")

    @property
    def capacity(self) -> int:
        return self.engine.n_new - 1


def _all_rows_engine(model, B: int, L: int, text_only: bool) -> OpenVLAEngine:
    # engines live on the model (LRU-bounded, PrismaticForConditionalGeneration._lru), never in a module global
    return model._lru(model._forward_engines, (B, L, text_only),
                      lambda: OpenVLAEngine(model.weights, B, L, all_rows=True, use_mask=True, text_only=text_only))


def forward_all_rows(model, input_ids: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor],
                     pixel_values: Optional[torch.Tensor], labels: Optional[torch.Tensor],
                     inputs_embeds: Optional[torch.Tensor] = None, output_hidden_states: bool = False):
    """Returns (loss or None, logits fp32 [B, S, vocab], projector features bf16 [B, 256, D] or None, hidden states or
    None). `pixel_values is None` is the reference's language-only branch (S = L); there `inputs_embeds` [B, L, D] may
    stand in for `input_ids` (an extension: the reference asserts it away, modeling_prismatic.py:345)."""
    dev = model.device
    text_only = pixel_values is None
    if inputs_embeds is not None and not text_only:
        raise ValueError("inputs_embeds replaces the token embeddings of the language-only forward; the multimodal forward "
                         "embeds input_ids itself (modeling_prismatic.py:380)")
    B, L = (inputs_embeds if input_ids is None else input_ids).shape[:2]
    eng = _all_rows_engine(model, B, L, text_only)
    P, S = eng.n_patches, eng.S
    if input_ids is not None:
        eng.set_inputs(input_ids.to(dev), None if text_only else pixel_values.to(dev))
    if attention_mask is None:
        eng.key_mask.fill_(1)
    else:
        m = attention_mask.to(dev).to(torch.uint8)
        eng.key_mask[:, :1] = m[:, :1]
        eng.key_mask[:, 1:1 + P] = 1
        eng.key_mask[:, 1 + P:] = m[:, 1:]
    proj = None
    if not text_only:
        ops.run_all(eng.vision_ops + eng.projector_ops)
        proj = eng.x[:, 1:1 + P].clone()          # rows 1..256 are overwritten in place by the decoder layers
    plan = eng.prefill_ops
    if inputs_embeds is not None:
        if tuple(inputs_embeds.shape) != (B, L, model.dims.llm_dim):
            raise ValueError(f"inputs_embeds must be [{B}, {L}, {model.dims.llm_dim}]")
        eng.epoch += 1
        eng.x.copy_(inputs_embeds.to(dev).to(torch.bfloat16))
    else:
        ops.run_all(plan[:1])                     # token-embedding gather / splice
    hidden: Optional[List[torch.Tensor]] = [eng.x.clone()] if output_hidden_states else None
    if output_hidden_states:                      # HF: embeddings, each layer's output, the last one after the final norm
        at = 1
        for li, end in enumerate(eng.layer_ends):
            ops.run_all(plan[at:end])
            at = end
            if li + 1 < len(eng.layer_ends):
                hidden.append(eng.x.clone())
        ops.run_all(plan[at:])
        hidden.append(eng.h.view(B, S, -1).clone())
    else:
        ops.run_all(plan[1:])
    logits = eng.logits_all.view(B, S, -1).clone()      # fresh tensor like the reference's: the engine buffer is reused
    loss = None
    if labels is not None:
        lab = labels.to(dev)
        full = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64, device=dev)
        full[:, :1] = lab[:, :1]
        full[:, 1 + P:] = lab[:, 1:]
        targets = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64, device=dev)
        targets[:, :-1] = full[:, 1:]          # position t predicts token t+1
        row_loss = torch.empty(B * S, dtype=torch.float32, device=dev)
        mean_cnt = torch.empty(2, dtype=torch.float32, device=dev)
        ops.cross_entropy(eng.logits_all, targets.view(-1), row_loss, mean_cnt, IGNORE_INDEX)
        loss = mean_cnt[0].clone()
    return loss, logits, proj, (tuple(hidden) if hidden is not None else None)


def forward_prefill_cached(model, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                           pixel_values: torch.Tensor, max_new_tokens: int) -> Tuple[torch.Tensor, EngineKVCache]:
    """Multimodal forward with `use_cache=True` = the first step of generation: runs exactly the launches `generate()`
    runs for vision → projector → prefill → first-token head (the last decoder layer and lm_head on the last position
    only, engine.py) and returns (logits [B, 1, vocab] of that position, cache handle). `logits[:, -1]` is what
    GenerationMixin reads; the other S-1 rows are what `logits_to_keep=1` drops in current transformers."""
    dev = model.device
    B, L = input_ids.shape
    padded = attention_mask is not None and not bool(attention_mask.bool().all())
    eng = model.engine(B, L, max_new_tokens, padded=padded, cached=True)
    if padded:
        eng.set_padded_inputs(input_ids.to(dev), pixel_values.to(dev), attention_mask)
    else:
        eng.set_inputs(input_ids.to(dev), pixel_values.to(dev))
    eng.run_vision()
    ops.run_all(eng.projector_ops + eng.prefill_ops)
    return eng.logits[0].clone().unsqueeze(1), EngineKVCache(eng)


def forward_cached_step(model, input_ids: torch.Tensor, cache: EngineKVCache) -> torch.Tensor:
    """Cached generation branch (modeling_prismatic.py:325-341; batch > 1 is an extension over its batch-1 assert): the
    token ids [B, 1] are appended at position S + steps, logits [B, 1, vocab] of that position come back. Same launches as
    decode step `steps + 1` of `generate()`, fed with the caller's token instead of the engine's own argmax."""
    if not isinstance(cache, EngineKVCache):
        raise TypeError("past_key_values must be the handle returned by a forward(..., use_cache=True) of this model")
    cache.check()
    eng = cache.engine
    t = cache.steps + 1
    if t > cache.capacity:
        raise RuntimeError(f"this KV cache was sized for {eng.n_new} new tokens (forward(..., use_cache=True) → "
                           f"model.cache_new_tokens); raise it before the prefill to generate more")
    if tuple(input_ids.shape) != (eng.B, 1):
        raise ValueError(f"cached generation step takes input_ids [{eng.B}, 1]")
    eng.gen_ids[t - 1].copy_(input_ids.to(model.device)[:, 0])
    ops.run_all(eng.decode_ops[t - 1])
    cache.steps = t
    return eng.logits[t].clone().unsqueeze(1)
