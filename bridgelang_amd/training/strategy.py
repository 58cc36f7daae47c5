"""Host side of VLA training behind the reference's strategy interface (prismatic/training/strategies/base_strategy.py
`TrainingStrategy`, fsdp.py `FSDPStrategy`; selected by `get_train_strategy`, prismatic/training/materialize.py):
same constructor arguments, `run_setup` / `clip_grad_norm` / `save_checkpoint` / `run_vla_training`, same metrics
(VLAMetrics: loss, action-token accuracy, L1, step time, lr → trackers) and the same checkpoint files. The device work
of every iteration is training/step.py::TrainStep; the FSDP wrapper, autocast and activation checkpointing have no
counterpart here (sharded optimizer over replicated bf16 weights, see sharding.py).
"""
from __future__ import annotations

import json
import math
import time
from collections import defaultdict, deque
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist

from .checkpoint import to_model_state_dicts
from .metrics import vla_action_metrics
from .step import STAGES, ParamStore, TrainStep


def _rank_world() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


# ---- learning-rate schedules (fsdp.py:195-236 → transformers get_cosine_schedule_with_warmup / get_constant_schedule) ----
def lr_at(step: int, base_lr: float, kind: str, num_training_steps: int, num_warmup_steps: int) -> float:
    """Learning rate used by optimizer step number `step` (0-based: the scheduler is stepped after the optimizer)."""
    if kind == "constant":
        return base_lr
    if kind == "linear-warmup+cosine-decay":
        if step < num_warmup_steps:
            return base_lr * step / max(1, num_warmup_steps)
        progress = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
    raise ValueError(f"Learning Rate Schedule with type `{kind}` is not supported!")


# ---- metrics (prismatic/training/metrics.py:208-350) -------------------------------------------------------------------
class JSONLinesTracker:
    def __init__(self, run_id: str, run_dir: Path, hparams: Dict[str, Any]):
        self.run_id, self.run_dir, self.hparams = run_id, Path(run_dir), hparams

    def write_hyperparameters(self) -> None:
        self.run_dir.mkdir(parents=True, exist_ok=True)
        with open(self.run_dir / "run-metrics.jsonl", "a") as f:
            f.write(json.dumps({"run_id": self.run_id, "hparams": self.hparams}, default=str) + "\n")

    def write(self, _: int, metrics: Dict[str, Union[int, float]]) -> None:
        with open(self.run_dir / f"{self.run_id}.jsonl", "a") as f:
            f.write(json.dumps(metrics) + "\n")

    def finalize(self) -> None:
        return


class VLAMetrics:
    def __init__(self, active_trackers: Tuple[str, ...], run_id: str, run_dir: Path, hparams: Dict[str, Any],
                 grad_accumulation_steps: int = 1, window_size: int = 1, resume_step: Optional[int] = None,
                 resume_epoch: Optional[int] = None, **_: Any) -> None:
        self.run_id, self.run_dir, self.hparams = run_id, run_dir, hparams
        self.trackers = []
        for tracker_type in active_trackers:
            if tracker_type != "jsonl":      # the reference's other tracker is wandb (network service): out of scope
                raise ValueError(f"Tracker with type `{tracker_type} is not supported!")
            tracker = JSONLinesTracker(run_id, run_dir, hparams)
            tracker.write_hyperparameters()
            self.trackers.append(tracker)
        self.global_step = 0 if resume_step is None else resume_step
        self.epoch = 0 if resume_epoch is None else resume_epoch
        self.start_time, self.step_start_time = time.time(), time.time()
        self.state = {"loss_raw": deque(maxlen=grad_accumulation_steps), "loss": deque(maxlen=window_size),
                      "l1_loss": deque(maxlen=window_size), "action_accuracy": deque(maxlen=window_size),
                      "step_time": deque(maxlen=window_size), "lr": []}
        self.dataset_trackers = defaultdict(lambda: VLAMetrics((), "", "", {}))

    def log(self, global_step: int, metrics: Dict[str, Union[int, float]]) -> None:
        for tracker in self.trackers:
            tracker.write(global_step, metrics)

    def get_status(self, loss: Optional[float] = None) -> str:
        lr = self.state["lr"][-1] if len(self.state["lr"]) > 0 else 0
        base = f"=>> [Epoch {self.epoch:03d}] Global Step {self.global_step:06d} =>> LR :: {lr:.6f}"
        return base if loss is None else base + f" - Loss :: {loss:.4f}"

    def commit(self, *, global_step: Optional[int] = None, epoch: Optional[int] = None, lr: Optional[float] = None,
               update_step_time: bool = False, **kwargs) -> None:
        if global_step is not None:
            self.global_step = global_step
        if epoch is not None:
            self.epoch = epoch
        if _rank_world()[0] != 0:
            return
        if lr is not None:
            self.state["lr"].append(lr)
        if update_step_time:
            self.state["step_time"].append(time.time() - self.step_start_time)
            self.step_start_time = time.time()
        for key, value in kwargs.items():
            value = float(value)
            if key == "loss":
                self.state["loss_raw"].append(value)
                self.state["loss"].append(value)
            else:
                self.state[key].append(value)

    def commit_for_dataset(self, dataset_name: str, **kwargs) -> None:
        self.dataset_trackers[dataset_name].commit(**kwargs)

    def push(self) -> str:
        if _rank_world()[0] != 0:
            return ""
        mean = lambda k, st=self.state: float(np.mean(list(st[k]))) if len(st[k]) else float("nan")
        loss = mean("loss")
        dataset_metrics = {}
        for ds, tracker in self.dataset_trackers.items():
            dataset_metrics[f"{ds}/L1 Loss"] = mean("l1_loss", tracker.state)
            dataset_metrics[f"{ds}/Action Token Accuracy"] = mean("action_accuracy", tracker.state)
        prefix = "VLA Train"
        self.log(self.global_step, metrics={
            f"{prefix}/Step": self.global_step, f"{prefix}/Epoch": self.epoch, f"{prefix}/Loss": loss,
            f"{prefix}/L1 Loss": mean("l1_loss"), f"{prefix}/Action Token Accuracy": mean("action_accuracy"),
            f"{prefix}/Loss (Raw)": mean("loss_raw"), f"{prefix}/Learning Rate": self.state["lr"][-1],
            f"{prefix}/Step Time": mean("step_time"), **dataset_metrics})
        return self.get_status(loss)

    def finalize(self) -> None:
        for tracker in self.trackers:
            tracker.finalize()


# ---- strategy ----------------------------------------------------------------------------------------------------------
class ShardedOptimizerStrategy:
    """Drop-in for FSDPStrategy on the VLA path. `vlm` is the HF-interface model (extern/hf/modeling_prismatic.py)."""
    ALL_MODULE_KEYS = ("vision_backbone", "llm_backbone", "projector")

    def __init__(self, vlm, device_id: int, stage: str, epochs: int, max_steps: Optional[int], global_batch_size: int,
                 per_device_batch_size: int, learning_rate: float, weight_decay: float, max_grad_norm: float,
                 lr_scheduler_type: str, warmup_ratio: float, enable_gradient_checkpointing: bool = True,
                 enable_mixed_precision_training: bool = True, reduce_in_full_precision: bool = False,
                 mixed_precision_dtype: torch.dtype = torch.bfloat16, worker_init_fn: Optional[Callable[[int], None]] = None,
                 sharding_strategy: str = "shard-grad-op", max_text_len: int = 48,
                 recompute_activations: Optional[bool] = None, fp8_gemms: bool = False, fp8_wgrad: bool = False,
                 **_: Any) -> None:
        if stage not in STAGES:
            raise ValueError(f"Stage `{stage}` is not supported")
        if sharding_strategy not in ("shard-grad-op", "full-shard"):
            raise ValueError(f"FSDP Sharding Strategy {sharding_strategy} is not supported!")
        self.vlm, self.device_id, self.stage = vlm, device_id, stage
        self.epochs, self.max_steps = epochs, max_steps
        self.global_batch_size, self.per_device_batch_size = global_batch_size, per_device_batch_size
        self.learning_rate, self.weight_decay, self.max_grad_norm = learning_rate, weight_decay, max_grad_norm
        self.lr_scheduler_type, self.warmup_ratio = lr_scheduler_type, warmup_ratio
        self.reduce_in_full_precision = reduce_in_full_precision
        self.worker_init_fn = worker_init_fn
        assert mixed_precision_dtype == torch.bfloat16 and enable_mixed_precision_training, "the HIP path computes in bf16"
        self.rank, self.world = _rank_world()
        assert global_batch_size % per_device_batch_size == 0, "Per-device batch size must evenly divide global batch size!"
        self.grad_accumulation_steps = global_batch_size // per_device_batch_size // self.world
        vision, proj, llm = STAGES[stage]
        self.trainable_module_keys = [k for k, on in (("vision_backbone", vision), ("projector", proj),
                                                      ("llm_backbone", llm != "none")) if on]
        self.all_module_keys = list(self.ALL_MODULE_KEYS)
        self.max_text_len = max_text_len
        # "full-shard" (FSDP FULL_SHARD, fsdp.py:84-87): the decoder layers' parameters are sharded over the ranks and
        # gathered per layer around their use (TrainStep(shard_params=True)) in the stages that train the whole LLM; the
        # other stages, and "shard-grad-op", keep the bf16 weights replicated and shard gradients + optimizer state
        self.shard_params = sharding_strategy == "full-shard" and STAGES[stage][2] == "all"
        self.fp8_gemms = fp8_gemms      # extension (no reference flag): decoder-layer forward / dgrad GEMMs in e4m3
        self.fp8_wgrad = fp8_wgrad and fp8_gemms     # … and their weight-gradient GEMMs too (TrainStep(fp8_wgrad=True))
        # the reference checkpoints every decoder layer when `enable_gradient_checkpointing` (fsdp.py:171-183) because
        # 80 GB parts cannot keep the activations; here they stay resident unless they would not fit (None = decide from
        # free HBM when the step is planned), and True / False force either form
        self.enable_gradient_checkpointing = enable_gradient_checkpointing
        self.recompute_activations = recompute_activations
        self.store: Optional[ParamStore] = None
        self.step_engine: Optional[TrainStep] = None
        self.num_training_steps = self.num_warmup_steps = 0

    # -- setup --
    def run_setup(self, run_dir: Path, n_train_examples: int) -> None:
        n_train_examples = math.ceil(n_train_examples / self.global_batch_size) * self.global_batch_size
        self.num_training_steps = ((n_train_examples * self.epochs) // self.global_batch_size
                                   if self.max_steps is None else self.max_steps)
        lr_at(0, self.learning_rate, self.lr_scheduler_type, 1, 0)          # validates the schedule name
        self.num_warmup_steps = (int(self.num_training_steps * self.warmup_ratio)
                                 if self.lr_scheduler_type == "linear-warmup+cosine-decay" else 0)
        if self.shard_params:           # inference engines cached on the model hold the decoder-layer weights: drop them
            for cache in ("_engines", "_forward_engines"):
                getattr(self.vlm, cache, {}).clear()
        self.store = ParamStore(self.vlm.weights, self.stage, self.world, self.rank, shard_params=self.shard_params,
                                defer_grads=True)      # gradient buffers come with the first TrainStep (after it frees the layers)
        self._ensure_engine(self.max_text_len)

    def _ensure_engine(self, text_len: int) -> TrainStep:
        """(Re)plan the static step when a batch is longer than the planned text length; optimizer state lives in the
        ParamStore and survives."""
        if self.step_engine is None or text_len > self.step_engine.L:
            L = (text_len + 15) // 16 * 16
            self.step_engine = None
            torch.cuda.empty_cache()
            self.step_engine = TrainStep(
                self.vlm.weights, self.stage, self.per_device_batch_size, L, max_grad_norm=self.max_grad_norm,
                weight_decay=self.weight_decay, store=self.store, recompute=self._want_recompute(L),
                shard_params=self.shard_params, fp8=self.fp8_gemms, fp8_wgrad=self.fp8_wgrad, world=self.world, rank=self.rank,
                reduce_dtype=torch.float32 if self.reduce_in_full_precision else torch.bfloat16)
        return self.step_engine

    def _want_recompute(self, L: int) -> bool:
        if self.recompute_activations is not None:
            return bool(self.recompute_activations) and self.enable_gradient_checkpointing
        if not self.enable_gradient_checkpointing:
            return False
        d = self.vlm.weights.dims
        tokens = self.per_device_batch_size * (L + d.n_patches)
        saved = tokens * d.llm_layers * (8 * d.llm_dim + 3 * d.llm_inter) * 2        # bytes of per-layer activations
        free, _ = torch.cuda.mem_get_info(self.vlm.weights.embed.device)
        return saved > 0.8 * free

    def clip_grad_norm(self) -> torch.Tensor:
        return self.step_engine.clip_grad_norm()

    # -- checkpoints --
    def save_checkpoint(self, run_dir: Path, global_step: int, epoch: int, train_loss: Optional[float] = None,
                        only_trainable: bool = True) -> Optional[Path]:
        """`{"model": {module key: state dict}}` in the reference's native key names; trainable modules carry the fp32
        masters (gathered from the ranks' slices), frozen ones (only_trainable=False) the live bf16 weights."""
        masters = self.store.master_state_dict(self.step_engine.comm)       # collective: every rank takes part
        if self.rank != 0:
            return None
        hf = {} if only_trainable else {k: v.float().cpu() for k, v in self.vlm.weights.state_dict().items()}
        hf.update({k: v.cpu() for k, v in masters.items()})
        # never-executed tensors of a saved module (last ViT block, final norm, attention pool): the reference loads each
        # module's state dict strictly (prismatic.py:113-116), so they travel with it
        hf.update({k: v.float().cpu() for k, v in self.vlm.weights.passthrough.items() if k not in hf})
        keys = self.trainable_module_keys if only_trainable else self.all_module_keys
        model_state_dicts = to_model_state_dicts(hf, keys)
        checkpoint_dir = Path(run_dir) / "checkpoints"
        checkpoint_dir.mkdir(parents=True, exist_ok=True)
        tag = "inf" if train_loss is None else f"{train_loss:.4f}"
        path = checkpoint_dir / f"step-{global_step:06d}-epoch-{epoch:02d}-loss={tag}.pt"
        torch.save({"model": model_state_dicts}, path)
        return path

    # -- the loop (base_strategy.py:245-389) --
    def run_vla_training(self, vla_dataset, collator, action_tokenizer, metrics: VLAMetrics, save_interval: int = 2500,
                         save_full_model: bool = True) -> None:
        from torch.utils.data import DataLoader, IterableDataset
        assert isinstance(vla_dataset, IterableDataset), "VLA training expects an IterableDataset!"
        assert self.grad_accumulation_steps == 1, "VLA training does not support gradient accumulation!"
        dataloader = DataLoader(vla_dataset, batch_size=self.per_device_batch_size, sampler=None, collate_fn=collator,
                                num_workers=0, worker_init_fn=self.worker_init_fn)
        num_patches = self.vlm.weights.dims.n_patches
        for batch in dataloader:
            ids = batch["input_ids"]
            if ids.shape[0] != self.per_device_batch_size:
                continue                                                    # ragged tail of a finite iterable
            eng = self._ensure_engine(ids.shape[1])
            pv = batch["pixel_values"]
            if isinstance(pv, dict):          # the native fused backbone's transform yields {"dino", "siglip"} (dinosiglip_vit.py:33-40)
                pv = torch.cat([pv["dino"], pv["siglip"]], dim=1)
            eng.set_batch(ids, batch["attention_mask"], pv, batch["labels"])
            loss = eng.forward(graph=True)                    # static plans replayed as HIP graphs
            metrics.commit(loss=loss)
            eng.backward(graph=True)                          # (sharded runs keep the backward eager: per-bucket collectives)
            # action metrics (base_strategy.py:314-329) on the positions this batch really has
            S_b = num_patches + ids.shape[1]
            logits = eng.logits.view(eng.B, eng.S, -1)[:, :S_b]
            m = vla_action_metrics(logits, batch["labels"], action_tokenizer, num_patches=num_patches)
            metrics.commit(action_accuracy=m["action_accuracy"], l1_loss=m["l1_loss"], update_step_time=True)
            if self.rank == 0 and "dataset_names" in batch:
                datasets = set(batch["dataset_names"])
                if len(datasets) > 1:
                    for ds in datasets:
                        sel = torch.tensor([elem == ds for elem in batch["dataset_names"]])
                        md = vla_action_metrics(logits[sel.to(logits.device)], batch["labels"][sel], action_tokenizer,
                                                num_patches=num_patches)
                        metrics.commit_for_dataset(dataset_name=ds.decode() if isinstance(ds, bytes) else str(ds),
                                                   action_accuracy=md["action_accuracy"], l1_loss=md["l1_loss"])
            self.clip_grad_norm()
            lr = lr_at(metrics.global_step, self.learning_rate, self.lr_scheduler_type, self.num_training_steps,
                       self.num_warmup_steps)
            eng.optimizer_step(lr, graph=True)
            try:
                epoch = (metrics.global_step + 1) // (len(vla_dataset) // self.global_batch_size)
            except (TypeError, ZeroDivisionError):
                epoch = 0
            next_lr = lr_at(metrics.global_step + 1, self.learning_rate, self.lr_scheduler_type, self.num_training_steps,
                            self.num_warmup_steps)
            metrics.commit(global_step=metrics.global_step + 1, epoch=epoch, lr=next_lr)   # get_last_lr() after .step()
            metrics.push()
            terminate = self.max_steps is not None and metrics.global_step >= self.max_steps
            if terminate or (metrics.global_step % save_interval) == 0:
                self.save_checkpoint(metrics.run_dir, metrics.global_step, epoch, float(loss),
                                     only_trainable=not save_full_model)
                if dist.is_initialized():
                    dist.barrier()
                if terminate:
                    self.finish()
                    return
        self.finish()

    def finish(self) -> None:
        """End of training: a parameter-sharded run gathers the decoder layers back into the model's own allocation, so
        `vlm` is whole again for inference / `save_pretrained` (collective: every rank calls it)."""
        if self.step_engine is not None:
            self.step_engine.materialize_params()


def get_train_strategy(train_strategy: str, **kwargs) -> ShardedOptimizerStrategy:
    """prismatic/training/materialize.py:get_train_strategy: "fsdp-shard-grad-op" shards gradients + optimizer state under
    replicated bf16 weights; "fsdp-full-shard" additionally shards the decoder layers' parameters (gathered per layer)."""
    if train_strategy not in ("fsdp-shard-grad-op", "fsdp-full-shard"):
        raise ValueError(f"Train Strategy `{train_strategy}` is not supported!")
    return ShardedOptimizerStrategy(sharding_strategy=train_strategy[len("fsdp-"):], **kwargs)
