"""LoRA / full fine-tuning loop of vla-scripts/finetune.py:118-350 on the HIP training step: same config fields, same
per-step metrics (loss, action-token accuracy, L1 — smoothed over the gradient-accumulation window), AdamW at a constant
learning rate, adapter weights saved under their PEFT names plus the merged model weights (`merge_and_unload`)."""
from __future__ import annotations

import json
from collections import deque
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Any, Dict, Iterable, Optional

import torch.distributed as dist

from .lora import LoraAdapters
from .metrics import vla_action_metrics
from .step import TrainStep


@dataclass
class FinetuneConfig:
    """The fields of the reference's FinetuneConfig (finetune.py:74-110) that steer computation; dataset / W&B / hub
    fields belong to the caller."""
    run_root_dir: Path = Path("runs")
    adapter_tmp_dir: Path = Path("adapter-tmp")
    batch_size: int = 16
    max_steps: int = 200_000
    save_steps: int = 5000
    learning_rate: float = 5e-4
    grad_accumulation_steps: int = 1
    save_latest_checkpoint_only: bool = True
    use_lora: bool = True
    lora_rank: int = 32
    lora_dropout: float = 0.0
    use_quantization: bool = False
    max_text_len: int = 48
    log_every: int = 10


def finetune(vlm, dataloader: Iterable[Dict[str, Any]], action_tokenizer, cfg: FinetuneConfig,
             log_path: Optional[Path] = None) -> Dict[str, Any]:
    """`vlm`: OpenVLAForActionPrediction (HIP); `dataloader` yields PaddedCollatorForActionPrediction batches of
    cfg.batch_size samples. Returns the last smoothed metrics and the paths written."""
    if cfg.use_quantization:
        raise NotImplementedError("4-bit base weights (bitsandbytes) are outside the MI355X path: 288 GB HBM holds bf16")
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    w = vlm.weights
    lora = LoraAdapters(w, r=cfg.lora_rank, dropout=cfg.lora_dropout) if cfg.use_lora else None
    stage = "lora" if cfg.use_lora else "vla-full-train"
    engine: Optional[TrainStep] = None
    store = None
    num_patches = w.dims.n_patches
    recent = {k: deque(maxlen=cfg.grad_accumulation_steps) for k in ("loss", "acc", "l1")}
    out: Dict[str, Any] = {"steps": 0}
    log = open(log_path, "a") if (log_path is not None and rank == 0) else None
    for batch_idx, batch in enumerate(dataloader):
        ids = batch["input_ids"]
        if engine is None or ids.shape[1] > engine.L:
            L = (max(ids.shape[1], cfg.max_text_len) + 15) // 16 * 16
            engine = TrainStep(w, stage, cfg.batch_size, L, lora=lora, store=store, world=world, rank=rank,
                               max_grad_norm=float("inf"), weight_decay=0.01)      # AdamW(params, lr): torch defaults
            store = engine.store
        engine.set_batch(ids, batch["attention_mask"], batch["pixel_values"], batch["labels"])
        loss = engine.forward(graph=True)
        engine.backward(graph=True)
        logits = engine.logits.view(engine.B, engine.S, -1)[:, :num_patches + ids.shape[1]]
        m = vla_action_metrics(logits, batch["labels"], action_tokenizer, num_patches=num_patches)
        recent["loss"].append(float(loss)); recent["acc"].append(m["action_accuracy"]); recent["l1"].append(m["l1_loss"])
        accum = cfg.grad_accumulation_steps
        if accum > 1:                                  # normalized_loss = loss / accum (finetune.py:256-262); under the sharded
            engine.accumulate(1.0 / accum)             # optimizer every micro-batch is reduced and this rank's slices accumulate
        if (batch_idx + 1) % accum == 0:               # finetune.py:307-310
            if accum > 1:
                engine.use_accumulated()
            engine.clip_grad_norm()                    # max_norm = inf: only feeds the (unit) coefficient AdamW reads
            engine.optimizer_step(cfg.learning_rate, graph=True)
        step = batch_idx // accum                      # gradient step index (finetune.py:288)
        window_done = (batch_idx + 1) % accum == 0
        done_steps = (batch_idx + 1) // accum
        sm = {k: sum(v) / len(v) for k, v in recent.items()}
        out.update(steps=done_steps, train_loss=sm["loss"], action_accuracy=sm["acc"], l1_loss=sm["l1"])
        if log is not None and window_done and step % cfg.log_every == 0:
            log.write(json.dumps({"step": step, "train_loss": sm["loss"], "action_accuracy": sm["acc"], "l1_loss": sm["l1"]}) + "\n")
            log.flush()
        if window_done and ((step > 0 and step % cfg.save_steps == 0) or done_steps == cfg.max_steps):
            out.update(save_checkpoint(vlm, lora, engine, cfg, step, rank))
        if done_steps >= cfg.max_steps:
            break
    if log is not None:
        log.close()
    return out


def save_checkpoint(vlm, lora: Optional[LoraAdapters], engine: TrainStep, cfg: FinetuneConfig, step: int, rank: int) -> Dict[str, str]:
    """finetune.py:318-350: adapter weights (PEFT names, fp32 masters) + the merged bf16 model."""
    from safetensors.torch import save_file
    full = engine.store.full_master(engine.comm)           # collective
    if rank != 0:
        return {}
    run_dir = Path(cfg.run_root_dir) if cfg.save_latest_checkpoint_only else Path(cfg.run_root_dir) / f"step-{step}"
    run_dir.mkdir(parents=True, exist_ok=True)
    paths = {}
    if lora is not None:
        adapter_dir = Path(cfg.adapter_tmp_dir)
        adapter_dir.mkdir(parents=True, exist_ok=True)
        masters = {u.key: full[u.offset:u.offset + u.numel] for u in engine.store.units}
        save_file({k: v.contiguous() for k, v in lora.state_dict(masters).items()}, str(adapter_dir / "adapter_model.safetensors"))
        (adapter_dir / "adapter_config.json").write_text(json.dumps({
            "peft_type": "LORA", "r": lora.r, "lora_alpha": lora.alpha, "lora_dropout": 0.0, "target_modules": "all-linear",
            "init_lora_weights": "gaussian"}))
        merged = lora.merged_state_dict()
        paths["adapter"] = str(adapter_dir / "adapter_model.safetensors")
    else:
        merged = vlm.state_dict()
    save_file({k: v.contiguous().cpu() for k, v in merged.items()}, str(run_dir / "model.safetensors"))
    (run_dir / "finetune_config.json").write_text(json.dumps({k: str(v) for k, v in asdict(cfg).items()}))
    paths["model"] = str(run_dir / "model.safetensors")
    return paths
