"""vla-scripts/finetune.py on MI355X — LoRA (default r = 32) or full fine-tuning of an OpenVLA checkpoint.

Same command line as the reference's script (vla-scripts/finetune.py:74-110; README.md:161-197):

    torchrun --standalone --nnodes 1 --nproc-per-node 1 vla-scripts/finetune.py \
        --vla_path <local openvla-7b dir> --data_root_dir <DIR> --dataset_name bridge_orig --run_root_dir <DIR> \
        --adapter_tmp_dir <DIR> --lora_rank 32 --batch_size 16 --grad_accumulation_steps 1 --learning_rate 5e-4 \
        --image_aug False --save_steps 5000

Every `FinetuneConfig` field of the reference is a flag with the reference's default. The reference wraps the HF model in
PEFT + DDP and lets autograd do the rest (finetune.py:174-189); here the adapters, the backward pass and AdamW are the
hand-written training step (bridgelang_amd/training/{lora,step}.py) and multi-GPU runs shard the optimizer state over RCCL.
Nothing is downloaded: `--vla_path` is a local HF export, or `synthetic:openvla-7b` / `synthetic:openvla-tiny` for the
seeded synthetic checkpoint; `--dataset_name dummy` trains on DummyDataset (the RLDS reader is outside the hot path);
W&B logging is replaced by a JSONL file in the run directory with the same three keys (train_loss, action_accuracy,
l1_loss).
"""
from __future__ import annotations

import json
import sys
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from bridgelang_amd.conf import cli  # noqa: E402


@dataclass
class FinetuneConfig:
    # fmt: off
    vla_path: str = "openvla/openvla-7b"
    data_root_dir: Path = Path("datasets/open-x-embodiment")
    dataset_name: str = "droid_wipe"
    run_root_dir: Path = Path("runs")
    adapter_tmp_dir: Path = Path("adapter-tmp")
    batch_size: int = 16
    max_steps: int = 200_000
    save_steps: int = 5000
    learning_rate: float = 5e-4
    grad_accumulation_steps: int = 1
    image_aug: bool = True
    shuffle_buffer_size: int = 100_000
    save_latest_checkpoint_only: bool = True
    use_lora: bool = True
    lora_rank: int = 32
    lora_dropout: float = 0.0
    use_quantization: bool = False
    wandb_project: str = "openvla"
    wandb_entity: str = "stanford-voltron"
    run_id_note: Optional[str] = None
    # -- additions (see module docstring) --
    tokenizer: Optional[str] = None          # default: the tokenizer files in --vla_path, else the synthetic stand-in
    dummy_length: int = 10000
    seed: int = 7
    # fmt: on


def experiment_id(cfg: FinetuneConfig) -> str:
    """finetune.py:125-137."""
    exp_id = (f"{cfg.vla_path.split('/')[-1]}+{cfg.dataset_name}+b{cfg.batch_size * cfg.grad_accumulation_steps}"
              f"+lr-{cfg.learning_rate}")
    if cfg.use_lora:
        exp_id += f"+lora-r{cfg.lora_rank}+dropout-{cfg.lora_dropout}"
    if cfg.use_quantization:
        exp_id += "+q-4bit"
    if cfg.run_id_note is not None:
        exp_id += f"--{cfg.run_id_note}"
    if cfg.image_aug:
        exp_id += "--image_aug"
    return exp_id


def finetune(cfg: FinetuneConfig) -> dict:
    import torch
    import torch.distributed as dist
    from bridgelang_amd import replicas, weights as W
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import OpenVLAForActionPrediction, register_auto_classes
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor, PrismaticProcessor
    from bridgelang_amd.training import finetune as F
    from bridgelang_amd.util.data_utils import PaddedCollatorForActionPrediction
    from bridgelang_amd.util.synthetic_tokenizer import load_tokenizer
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    from bridgelang_amd.vla.datasets import DummyDataset

    print(f"Fine-tuning OpenVLA Model `{cfg.vla_path}` on `{cfg.dataset_name}`")
    assert torch.cuda.is_available(), "Fine-tuning assumes at least one GPU is available!"
    rank, local_rank, world = replicas.env_rank()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    replicas.init("nccl", dev)
    torch.manual_seed(cfg.seed)
    exp_id = experiment_id(cfg)
    run_dir, adapter_dir = Path(cfg.run_root_dir) / exp_id, Path(cfg.adapter_tmp_dir) / exp_id
    run_dir.mkdir(parents=True, exist_ok=True)
    register_auto_classes()                                  # finetune.py:151-154
    if cfg.vla_path.startswith("synthetic:"):
        dims = {"openvla-7b": W.openvla_7b_dims, "openvla-tiny": W.tiny_dims}[cfg.vla_path.split(":", 1)[1]]()
        vla = OpenVLAForActionPrediction(OpenVLAConfig(norm_stats={}), device=dev, dims=dims).init_synthetic(seed=cfg.seed)
        tok_src = cfg.tokenizer or "synthetic"
    else:
        vla = OpenVLAForActionPrediction.from_pretrained(cfg.vla_path, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
                                                         trust_remote_code=True, device=dev)
        has_tok = any((Path(cfg.vla_path) / n).exists() for n in ("tokenizer.json", "tokenizer.model"))
        tok_src = cfg.tokenizer or (cfg.vla_path if has_tok else "synthetic")
    tok = load_tokenizer(tok_src)
    processor = PrismaticProcessor(PrismaticImageProcessor(), tok)
    at = ActionTokenizer(tok)
    if cfg.dataset_name != "dummy":
        raise NotImplementedError(f"dataset `{cfg.dataset_name}` needs the RLDS/TFDS reader (outside the hot path, SURVEY §2 "
                                  f"row 16); `--dataset_name dummy` trains on DummyDataset")
    ds = DummyDataset(at, tok, processor.image_processor.apply_transform, length=cfg.dummy_length, seed=cfg.seed + 1000 * rank)
    if rank == 0:
        (run_dir / "dataset_statistics.json").write_text(json.dumps(
            {k: {kk: {s: [float(x) for x in v] for s, v in vv.items()} for kk, vv in d.items()} for k, d in ds.dataset_statistics.items()},
            indent=2))
    collator = PaddedCollatorForActionPrediction(tok.model_max_length, tok.pad_token_id, padding_side="right")
    loader = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, sampler=None, collate_fn=collator, num_workers=0,
                                         drop_last=True)
    fc = F.FinetuneConfig(run_root_dir=run_dir, adapter_tmp_dir=adapter_dir, batch_size=cfg.batch_size, max_steps=cfg.max_steps,
                          save_steps=cfg.save_steps, learning_rate=cfg.learning_rate, grad_accumulation_steps=cfg.grad_accumulation_steps,
                          save_latest_checkpoint_only=cfg.save_latest_checkpoint_only, use_lora=cfg.use_lora, lora_rank=cfg.lora_rank,
                          lora_dropout=cfg.lora_dropout, use_quantization=cfg.use_quantization)
    out = F.finetune(vla, loader, at, fc, log_path=run_dir / "train_log.jsonl")
    if rank == 0:
        (run_dir / "finetune_args.json").write_text(json.dumps(cli.encode(cfg), indent=2))
        tok.save_pretrained(run_dir)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    out["run_dir"] = str(run_dir)
    return out


if __name__ == "__main__":
    finetune(cli.parse(FinetuneConfig))
