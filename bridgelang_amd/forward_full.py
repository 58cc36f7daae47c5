"""All-position multimodal forward (+ shifted cross-entropy) behind `PrismaticForConditionalGeneration.forward`.

Mirrors the reference's multimodal branch (modeling_prismatic.py:362-415): attention mask and labels get 256 columns
inserted after column 0 (True / -100), HF's CausalLM loss shifts labels by one and averages over non-ignored tokens
(SURVEY App. A.3). Host code here only builds index tensors; the arithmetic is the HIP plan in engine.py plus
bl_cross_entropy_f32.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops
from .engine import OpenVLAEngine

IGNORE_INDEX = -100


def forward_all_rows(model, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                     pixel_values: torch.Tensor, labels: Optional[torch.Tensor]):
    """Returns (loss or None, logits fp32 [B, S, vocab], projector features bf16 [B, 256, D])."""
    dev = model.device
    B, L = input_ids.shape
    # engines live on the model (LRU-bounded, PrismaticForConditionalGeneration._lru), never in a module global
    eng = model._lru(model._forward_engines, (B, L), lambda: OpenVLAEngine(model.weights, B, L, all_rows=True, use_mask=True))
    P, S = model.dims.n_patches, eng.S
    eng.set_inputs(input_ids.to(dev), pixel_values.to(dev))
    if attention_mask is None:
        eng.key_mask.fill_(1)
    else:
        m = attention_mask.to(dev).to(torch.uint8)
        eng.key_mask[:, :1] = m[:, :1]
        eng.key_mask[:, 1:1 + P] = 1
        eng.key_mask[:, 1 + P:] = m[:, 1:]
    ops.run_all(eng.vision_ops + eng.projector_ops)
    proj = eng.x[:, 1:1 + P].clone()          # rows 1..256 are overwritten in place by the decoder layers
    ops.run_all(eng.prefill_ops)
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
    return loss, logits, proj
