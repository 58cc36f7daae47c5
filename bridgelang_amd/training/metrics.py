"""Per-step action metrics of the VLA training loop (prismatic/training/strategies/base_strategy.py:314-329; the LoRA
script's twin at vla-scripts/finetune.py:270-286): greedy predictions over the text positions, token accuracy and
continuous L1 on the action tokens only (labels > action_token_begin_idx).

    action_preds = logits[:, num_patches:-1].argmax(dim=2)      action_gt = labels[:, 1:]
    mask         = action_gt > action_tokenizer.action_token_begin_idx

The argmax over 32064 fp32 logits per position runs on the device (bl_argmax_f32, ties → first index as torch.argmax);
the handful of resulting ids are compared on the host exactly as the reference does after its `.cpu()`.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .. import ops


def vla_action_metrics(logits: torch.Tensor, labels: torch.Tensor, action_tokenizer, num_patches: int = 256) -> Dict[str, float]:
    """logits fp32 [B, num_patches + L, vocab] (device), labels int64 [B, L] (unshifted, -100 = ignore)."""
    B, S, V = logits.shape
    preds = torch.empty(B * S, dtype=torch.int64, device=logits.device)
    ops.argmax(logits.reshape(B * S, V), preds)
    action_preds = preds.view(B, S)[:, num_patches:-1].cpu()
    action_gt = labels[:, 1:].cpu()
    mask = action_gt > action_tokenizer.action_token_begin_idx
    correct = (action_preds == action_gt) & mask
    n = int(mask.sum())
    acc = float(correct.sum()) / n if n else float("nan")
    cont_pred = action_tokenizer.decode_token_ids_to_actions(action_preds[mask].numpy())
    cont_gt = action_tokenizer.decode_token_ids_to_actions(action_gt[mask].numpy())
    l1 = float(np.abs(cont_pred - cont_gt).mean()) if n else float("nan")
    return {"action_accuracy": acc, "l1_loss": l1, "n_action_tokens": n}
