#!/usr/bin/env python3
"""Loss / gradient-norm trajectory of the 7B full fine-tune step with the decoder GEMMs in bf16 vs e4m3 (TrainStep(fp8=True))
on the same synthetic checkpoint and the same sequence of batches:  python tools/fp8_vs_bf16_curve.py [steps] > curve.json"""
import gc
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import weights as W
from bridgelang_amd.training.step import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
B, L = 8, 40


def batch(seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 31000, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
    ids[:, -1] = 2
    labels = torch.full((B, L), -100)
    labels[:, -8:] = ids[:, -8:]
    return ids, labels, torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)


out = {}
for fp8 in (False, True):
    w = W.allocate(W.openvla_7b_dims(), dev).fill_synthetic(seed=0)
    ts = TrainStep(w, "vla-train", B, L, max_grad_norm=1.0, weight_decay=0.0, fp8=fp8)
    log = []
    for s in range(steps):
        ids, labels, pv = batch(1000 + s % 4)                    # four batches, revisited: the loss must fall
        ts.set_batch(ids, None, pv, labels)
        loss, norm = ts.step(2e-5)
        log.append((round(loss.item(), 5), round(norm.item(), 4)))
    out["fp8" if fp8 else "bf16"] = log
    del ts, w
    gc.collect()
    torch.cuda.empty_cache()
rel = [abs(a[0] - b[0]) / abs(a[0]) for a, b in zip(out["bf16"], out["fp8"])]
out["max_rel_loss_diff"] = round(max(rel), 5)
out["config"] = f"openvla-7b vla-train, B={B} x S={L + 256}, lr 2e-5, clip 1.0, 4 batches revisited, {steps} steps"
print(json.dumps(out))
