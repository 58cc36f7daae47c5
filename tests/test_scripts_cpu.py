"""CPU: the entry scripts keep the reference's command lines (vla-scripts/train.py:50-103, finetune.py:74-110): the
documented invocations (README.md:184-197, 276-288) parse into the same field values, `--vla.type` picks the registered
VLAConfig, the run directory gets config.yaml + config.json, and the `prismatic` import root resolves to this package."""
import importlib.util
import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _load(name):
    path = ROOT / "vla-scripts" / f"{name}.py"
    spec = importlib.util.spec_from_file_location(f"vla_scripts_{name}", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_train_command_line_of_the_readme(tmp_path, monkeypatch):
    from bridgelang_amd.conf import VLAConfig, VLARegistry, cli
    T = _load("train")
    monkeypatch.setenv("WORLD_SIZE", "8")
    argv = ["--pretrained_checkpoint", "/ckpt/checkpoints/step-295000-epoch-40-loss=0.2200.pt",
            "--vla.type", "prism-dinosiglip-224px+mx-bridge", "--data_root_dir", "/data", "--run_root_dir", str(tmp_path),
            "--run_id", "myrun", "--image_aug", "False", "--wandb_project", "p", "--wandb_entity", "e",
            "--save_interval", "1000", "--is_resume", "False"]
    cfg = cli.parse(T.TrainConfig, argv)
    assert cfg.vla.vla_id == "prism-dinosiglip-224px+mx-bridge" and type(cfg.vla) is VLARegistry.DINOSIGLIP_224PX_MX_BRIDGE.value
    assert cfg.pretrained_checkpoint == Path("/ckpt/checkpoints/step-295000-epoch-40-loss=0.2200.pt")
    assert cfg.is_resume is False and cfg.image_aug is False and cfg.save_interval == 1000 and cfg.run_id == "myrun"
    # lifted optimisation parameters (train.py:83-98) and the reference's defaults for this config (conf/vla.py:62-91)
    assert (cfg.global_batch_size, cfg.per_device_batch_size, cfg.learning_rate, cfg.max_grad_norm) == (256, 32, 2e-5, 1.0)
    assert cfg.train_strategy == "fsdp-full-shard" and cfg.lr_scheduler_type == "constant" and cfg.vla.data_mix == "bridge"
    assert T.stage_of(cfg.vla) == "vla-full-train"
    # nested overrides + the default registry choice (OXE magic soup++: 64 GPUs)
    monkeypatch.setenv("WORLD_SIZE", "64")
    d = cli.parse(T.TrainConfig, ["--vla.learning_rate", "1e-4", "--vla.freeze_vision_backbone", "True", "--resume_step=10000"])
    assert d.vla.vla_id == "prism-dinosiglip-224px+mx-oxe-magic-soup-plus" and d.learning_rate == 1e-4 and d.resume_step == 10000
    assert T.stage_of(d.vla) == "vla-train"
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(AssertionError, match="Expected World Size"):
        cli.parse(T.TrainConfig, ["--vla.type", "prism-dinosiglip-224px+mx-bridge"])
    with pytest.raises(KeyError):
        cli.parse(T.TrainConfig, ["--vla.type", "siglip-224px+mx-bridge"])          # not on the MI355X path
    with pytest.raises(SystemExit):
        cli.parse(T.TrainConfig, ["--no_such_flag", "1"])
    # config dump: yaml + json, re-readable, registry choice recorded as `type`
    cli.dump_yaml_and_json(cfg, tmp_path)
    tree = json.loads((tmp_path / "config.json").read_text())
    assert tree["vla"]["type"] == "prism-dinosiglip-224px+mx-bridge" and tree["vla"]["base_vlm"] == "prism-dinosiglip-224px+7b"
    assert tree["run_root_dir"] == str(tmp_path) and (tmp_path / "config.yaml").exists()
    assert set(VLAConfig.get_known_choices()) >= {"prism-dinosiglip-224px+mx-bridge", "prism-dinosiglip-224px+mx-oxe-magic-soup-plus"}


def test_finetune_command_line_of_the_readme():
    from bridgelang_amd.conf import cli
    Fm = _load("finetune")
    argv = ["--vla_path", "openvla/openvla-7b", "--data_root_dir", "/data", "--dataset_name", "bridge_orig", "--run_root_dir", "/runs",
            "--adapter_tmp_dir", "/tmp/ad", "--lora_rank", "32", "--batch_size", "16", "--grad_accumulation_steps", "1",
            "--learning_rate", "5e-4", "--image_aug", "False", "--wandb_project", "p", "--wandb_entity", "e", "--save_steps", "5000"]
    cfg = cli.parse(Fm.FinetuneConfig, argv)
    assert (cfg.lora_rank, cfg.batch_size, cfg.learning_rate, cfg.save_steps, cfg.image_aug) == (32, 16, 5e-4, 5000, False)
    assert cfg.use_lora is True and cfg.lora_dropout == 0.0 and cfg.max_steps == 200_000 and cfg.adapter_tmp_dir == Path("/tmp/ad")
    assert Fm.experiment_id(cfg) == "openvla-7b+bridge_orig+b16+lr-0.0005+lora-r32+dropout-0.0"
    d = cli.parse(Fm.FinetuneConfig, ["--image_aug==False", "--use_lora", "false"])          # the README's `--flag==value` prose form
    assert d.image_aug is False and d.use_lora is False


def test_prismatic_import_root_resolves_here():
    """`from prismatic.… import …` lines of the reference's callers work unchanged (run_openvla_demo.py, finetune.py:36-46)."""
    from prismatic.extern.hf.configuration_prismatic import OpenVLAConfig
    from prismatic.extern.hf.modeling_prismatic import OpenVLAForActionPrediction
    from prismatic.extern.hf.processing_prismatic import PrismaticImageProcessor, PrismaticProcessor
    from prismatic.models.backbones.llm.prompting import PurePromptBuilder
    from prismatic.util.data_utils import PaddedCollatorForActionPrediction
    from prismatic.vla.action_tokenizer import ActionTokenizer
    from prismatic.vla.datasets import DummyDataset, RLDSBatchTransform
    from prismatic.conf import VLAConfig, VLARegistry
    from prismatic.models import load_vla
    from prismatic.training import VLAMetrics, get_train_strategy
    import bridgelang_amd.extern.hf.modeling_prismatic as M
    import bridgelang_amd.vla.action_tokenizer as A
    assert OpenVLAForActionPrediction is M.OpenVLAForActionPrediction and ActionTokenizer is A.ActionTokenizer
    from transformers import PreTrainedModel
    assert issubclass(OpenVLAForActionPrediction, PreTrainedModel) and OpenVLAForActionPrediction.config_class is OpenVLAConfig
    assert all(callable(x) for x in (PrismaticImageProcessor, PrismaticProcessor, PurePromptBuilder, PaddedCollatorForActionPrediction,
                                     DummyDataset, RLDSBatchTransform, load_vla, VLAMetrics, get_train_strategy))
    assert VLARegistry.DINOSIGLIP_224PX_MX_BRIDGE.vla_id in VLAConfig.get_known_choices()


def test_auto_class_registration_dispatches_to_the_hip_model(tmp_path):
    """finetune.py:151-166's registration + `AutoModel….from_pretrained(local_dir, …)` reaches
    OpenVLAForActionPrediction.from_pretrained (checked without a GPU by intercepting the loader)."""
    import torch
    import transformers
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf import modeling_prismatic as M
    from bridgelang_amd.models import load as L
    M.register_auto_classes()
    M.register_auto_classes()                        # idempotent
    cfg = OpenVLAConfig(norm_stats={"bridge_orig": {"action": {"q01": [0.0] * 7, "q99": [1.0] * 7}}})
    cfg.save_pretrained(tmp_path)
    back = transformers.AutoConfig.from_pretrained(tmp_path)
    assert type(back) is OpenVLAConfig and back.text_config.vocab_size == 32064 and back.image_sizes == [224, 224]
    seen = {}
    orig = L.load_hf_directory
    L.load_hf_directory = lambda cls, path, **kw: seen.update(cls=cls, path=str(path), kw=kw) or "model"
    try:
        for name in ("AutoModelForVision2Seq", "AutoModelForImageTextToText"):
            auto = getattr(transformers, name, None)
            if auto is None:
                continue
            out = auto.from_pretrained(tmp_path, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, trust_remote_code=True,
                                       attn_implementation="sdpa")
            assert out == "model" and seen["cls"] is M.OpenVLAForActionPrediction and seen["path"] == str(tmp_path)
            assert isinstance(seen["kw"]["config"], OpenVLAConfig)
        assert seen, "no HF Auto class for image+text → text models found"
        with pytest.raises(NotImplementedError):
            M.OpenVLAForActionPrediction.from_pretrained(tmp_path, torch_dtype=torch.float32)
    finally:
        L.load_hf_directory = orig
