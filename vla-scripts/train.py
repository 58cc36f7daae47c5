"""vla-scripts/train.py on MI355X — full / partial fine-tuning of an OpenVLA policy with the sharded training strategy.

Same command line as the reference's script (vla-scripts/train.py:50-138; README.md:224-262):

    torchrun --standalone --nnodes 1 --nproc-per-node 8 vla-scripts/train.py \
        --pretrained_checkpoint <.../checkpoints/step-295000-epoch-40-loss=0.2200.pt> \
        --vla.type prism-dinosiglip-224px+mx-bridge --data_root_dir <DIR> --run_root_dir <DIR> \
        --run_id <ID> --image_aug False --save_interval 2500 --is_resume False

Every `TrainConfig` / `VLAConfig` field of the reference is a flag with the reference's default. What differs, because
of where this runs: (i) nothing is downloaded — `--vla.base_vlm` must be a local run directory unless
`--pretrained_checkpoint` is given, and `--synthetic_init True` builds the named architecture with the seeded synthetic
checkpoint (benchmark / smoke runs); (ii) the RLDS/TFDS reader is outside the hot path: `--vla.data_mix
dummy` trains on `DummyDataset` (vla/datasets.py), any other mixture needs an RLDS reader and raises; (iii) trackers: `jsonl`
(W&B is a network service); (iv) `--tokenizer synthetic|<dir>` names the tokenizer (none ships offline); (v) `--fp8_gemms True`
runs the decoder layers' forward / input-gradient GEMMs on the e4m3 MFMA path (an extension; BASELINE configs[4]).
`--vla.train_strategy fsdp-full-shard` shards the decoder layers' parameters, `fsdp-shard-grad-op` keeps them replicated.
One process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE); collectives are RCCL over xGMI.
"""
from __future__ import annotations

import json
import os
import re
import sys
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Tuple, Union

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from bridgelang_amd.conf import VLAConfig, VLARegistry, cli  # noqa: E402


@dataclass
class TrainConfig:
    # fmt: off
    vla: VLAConfig = field(default_factory=VLAConfig.get_choice_class(VLARegistry.DINOSIGLIP_224PX_MX_OXE_MAGIC_SOUP_PLUS.vla_id))
    data_root_dir: Path = Path("datasets/open-x-embodiment")
    run_root_dir: Path = Path("runs")
    pretrained_checkpoint: Optional[Path] = None
    is_resume: bool = True
    resume_step: Optional[int] = None
    resume_epoch: Optional[int] = None
    run_id: Optional[str] = None
    run_id_note: Optional[str] = None
    save_interval: int = 2500
    image_aug: bool = False
    seed: int = 7
    hf_token: Union[str, Path] = Path(".hf_token")
    trackers: Tuple[str, ...] = ("jsonl",)
    wandb_project: str = "openvla"
    wandb_entity: str = "stanford-voltron"
    # -- additions (see module docstring) --
    synthetic_init: bool = False
    tokenizer: str = "synthetic"
    dummy_length: int = 10000
    fp8_gemms: bool = False                        # decoder-layer forward / dgrad GEMMs on the e4m3 MFMA path (BASELINE configs[4])
    fp8_wgrad: bool = False                        # with fp8_gemms: the weight-gradient GEMMs of those layers in e4m3 as well
    # fmt: on

    def __post_init__(self) -> None:
        """Lift the optimisation parameters from `self.vla`; gate on `expected_world_size` (train.py:83-103)."""
        for k in ("epochs", "max_steps", "global_batch_size", "per_device_batch_size", "learning_rate", "weight_decay",
                  "max_grad_norm", "lr_scheduler_type", "warmup_ratio", "train_strategy"):
            setattr(self, k, getattr(self.vla, k))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        assert self.vla.expected_world_size == world, \
            f"Expected World Size = {self.vla.expected_world_size} but Found {world} GPUs!"


def stage_of(v: VLAConfig) -> str:
    """train.py:152-172."""
    if not v.freeze_vision_backbone and not v.freeze_llm_backbone:
        return "vla-full-train"
    if v.freeze_vision_backbone and not v.freeze_llm_backbone:
        return "vla-train"
    assert v.unfreeze_last_llm_layer, "You should unfreeze at least the last layer of your LLM!"
    return "vla-sandwich-train" if not v.freeze_vision_backbone else "vla-last-layer-train"


# base VLM id → (vision backbone id, LLM backbone id, reduced test widths?) for --synthetic_init
_BASE_VLMS = {"prism-dinosiglip-224px+7b": ("dinosiglip-vit-so-224px", "llama2-7b-pure", False),
              "prism-dinosiglip-224px+13b": ("dinosiglip-vit-so-224px", "llama2-13b-pure", False),
              "openvla-tiny": ("dinosiglip-vit-so-224px", "llama2-7b-pure", True)}


def train(cfg: TrainConfig) -> Path:
    import torch
    import torch.distributed as dist
    from bridgelang_amd import replicas
    from bridgelang_amd.models.load import load_vla
    from bridgelang_amd.models.materialize import get_llm_backbone_and_tokenizer, get_vision_backbone_and_transform
    from bridgelang_amd.models.vlms import OpenVLA
    from bridgelang_amd.training.strategy import VLAMetrics, get_train_strategy
    from bridgelang_amd.util.data_utils import PaddedCollatorForActionPrediction
    from bridgelang_amd.util.synthetic_tokenizer import load_tokenizer
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    from bridgelang_amd.vla.datasets import DummyDataset, EpochIterable

    rank, local_rank, world = replicas.env_rank()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    replicas.init("nccl", dev)
    vla_id = cfg.vla.vla_id
    if cfg.run_id is None:
        cfg.run_id = f"{vla_id}+n{cfg.vla.expected_world_size // 8}+b{cfg.per_device_batch_size}+x{cfg.seed}"
    if cfg.run_id_note is not None:
        cfg.run_id += f"--{cfg.run_id_note}"
    if cfg.image_aug:
        cfg.run_id += "--image_aug"
    torch.manual_seed(cfg.seed)
    run_dir = Path(cfg.run_root_dir) / cfg.run_id
    (run_dir / "checkpoints").mkdir(parents=True, exist_ok=True)
    if rank == 0:
        cli.dump_yaml_and_json(cfg, run_dir)               # config.yaml + config.json (train.py:134-138)
    tok = load_tokenizer(cfg.tokenizer)
    if cfg.pretrained_checkpoint is not None:
        if cfg.is_resume:
            assert int(re.search("step-(.+?)-", cfg.pretrained_checkpoint.name).group(1)) == cfg.resume_step
            assert int(re.search("epoch-(.+?)-", cfg.pretrained_checkpoint.name).group(1)) == cfg.resume_epoch
        vlm = load_vla(cfg.pretrained_checkpoint, load_for_training=True, tokenizer=tok, device=dev)
    elif Path(str(cfg.vla.base_vlm)).is_dir():
        vlm = load_vla(cfg.vla.base_vlm, load_for_training=True, tokenizer=tok, device=dev)
    elif cfg.synthetic_init and str(cfg.vla.base_vlm) in _BASE_VLMS:
        from bridgelang_amd import weights as W
        vid, lid, tiny = _BASE_VLMS[str(cfg.vla.base_vlm)]
        vb, _ = get_vision_backbone_and_transform(vid, "resize-naive")
        lb, _ = get_llm_backbone_and_tokenizer(lid, llm_max_length=2048, inference_mode=False, tokenizer=tok)
        vlm = OpenVLA(str(cfg.vla.base_vlm), vb, lb, norm_stats={}, action_tokenizer=ActionTokenizer(tok), device=dev,
                      dims=W.tiny_dims() if tiny else None)
        vlm.hf.init_synthetic(seed=cfg.seed)
    else:
        raise FileNotFoundError(f"base VLM `{cfg.vla.base_vlm}` is not a local run directory and nothing is fetched from "
                                f"the hub: pass --pretrained_checkpoint, a local --vla.base_vlm, or --synthetic_init True")
    stage = stage_of(cfg.vla)
    vlm.freeze_backbones(stage)
    at = ActionTokenizer(tok)
    if cfg.vla.data_mix != "dummy":
        raise NotImplementedError(f"data mixture `{cfg.vla.data_mix}` needs the RLDS/TFDS reader (outside the hot path, "
                                  f"SURVEY §2 row 16); `--vla.data_mix dummy` trains on DummyDataset")
    ds = DummyDataset(at, tok, vlm.vision_backbone.get_image_transform(), prompt_builder_fn=vlm.llm_backbone.prompt_builder_fn,
                      length=cfg.dummy_length, seed=cfg.seed + 1000 * rank)
    vla_dataset = EpochIterable(ds)
    collator = PaddedCollatorForActionPrediction(tok.model_max_length, tok.pad_token_id, padding_side="right")
    if rank == 0:                                           # save_dataset_statistics (train.py:196-197)
        (run_dir / "dataset_statistics.json").write_text(json.dumps(
            {k: {kk: {s: [float(x) for x in v] for s, v in vv.items()} for kk, vv in d.items()} for k, d in ds.dataset_statistics.items()},
            indent=2))
    strategy = get_train_strategy(
        train_strategy=cfg.train_strategy, vlm=vlm.hf, device_id=local_rank, stage=stage, epochs=cfg.epochs, max_steps=cfg.max_steps,
        global_batch_size=cfg.global_batch_size, per_device_batch_size=cfg.per_device_batch_size, learning_rate=cfg.learning_rate,
        weight_decay=cfg.weight_decay, max_grad_norm=cfg.max_grad_norm, lr_scheduler_type=cfg.lr_scheduler_type,
        warmup_ratio=cfg.warmup_ratio, enable_gradient_checkpointing=cfg.vla.enable_gradient_checkpointing,
        enable_mixed_precision_training=cfg.vla.enable_mixed_precision_training,
        reduce_in_full_precision=cfg.vla.reduce_in_full_precision, fp8_gemms=cfg.fp8_gemms, fp8_wgrad=cfg.fp8_wgrad)
    strategy.run_setup(run_dir=run_dir, n_train_examples=len(vla_dataset))
    metrics = VLAMetrics(tuple(t for t in cfg.trackers if t != "wandb"), cfg.run_id, run_dir, cli.encode(cfg),
                         resume_step=cfg.resume_step, resume_epoch=cfg.resume_epoch)
    strategy.run_vla_training(vla_dataset, collator, at, metrics, save_interval=cfg.save_interval)
    metrics.finalize()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return run_dir


if __name__ == "__main__":
    train(cli.parse(TrainConfig))
