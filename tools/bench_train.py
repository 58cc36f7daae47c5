"""Time the single-GPU training step (TrainStep) on synthetic data: per-phase HIP-event times and model FLOP/s.
    python tools/bench_train.py --batch 16 --len 32 --steps 5 [--tiny] [--graph] [--stage vla-train]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgelang_amd.training.step import TrainStep  # noqa: E402
from bridgelang_amd.weights import allocate, openvla_7b_dims, tiny_dims  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--len", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--stage", default="vla-train")
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = tiny_dims() if a.tiny else openvla_7b_dims()
    t0 = time.time()
    w = allocate(dims, dev).fill_synthetic(seed=0)
    lora = None
    if a.stage == "lora":
        from bridgelang_amd.training.lora import LoraAdapters
        lora = LoraAdapters(w, r=32)
    ts = TrainStep(w, a.stage, a.batch, a.len, max_grad_norm=1.0 if lora is None else float("inf"),
                   weight_decay=0.0 if lora is None else 0.01, lora=lora)
    torch.cuda.synchronize()
    print(f"setup {time.time() - t0:.1f}s; trainable {ts.store.n_params / 1e9:.3f} B params; "
          f"HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    g = torch.Generator().manual_seed(0)
    B, L = a.batch, a.len
    ids = torch.randint(3, 31000, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
    ids[:, -1] = 2
    labels = torch.full((B, L), -100)
    labels[:, -8:] = ids[:, -8:]
    pv = torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)
    ts.set_batch(ids, None, pv, labels)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    losses = []
    for _ in range(a.warmup):
        loss, norm = ts.step(2e-5, graph=a.graph)
        losses.append(loss.item())
    phases = {"forward": 0.0, "backward": 0.0, "clip+adamw+repack": 0.0}
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        e = [ev() for _ in range(4)]
        e[0].record(); loss = ts.forward(a.graph)
        e[1].record(); ts.backward(a.graph)
        e[2].record(); ts.clip_grad_norm(); ts.optimizer_step(2e-5, a.graph)
        e[3].record()
        torch.cuda.synchronize()
        losses.append(loss.item())
        for k, i in zip(phases, range(3)):
            phases[k] += e[i].elapsed_time(e[i + 1])
    wall = (time.time() - t0) / a.steps * 1e3
    for k in phases:
        phases[k] /= a.steps
    fl = sum(op.flops for op in ts.forward_ops + ts.backward_ops + ts.vision_forward_ops)
    if not ts.train_vision:
        fl += sum(op.flops for op in ts._vis.vision_ops)
    tokens = B * ts.S
    out = {"workload": f"{dims.name} {a.stage} B={B} S={ts.S}", "ms_per_step": wall, "phases_ms": phases,
           "samples_per_s": B / wall * 1e3, "tokens_per_s": tokens / wall * 1e3, "tflop_per_step": fl / 1e12,
           "tflops": fl / wall / 1e9, "graph": a.graph, "losses": [round(x, 4) for x in losses],
           "hbm_gib": torch.cuda.max_memory_allocated() / 2**30}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
