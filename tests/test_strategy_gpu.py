"""The reference's VLA training loop surface (strategy.run_setup / run_vla_training / save_checkpoint, VLAMetrics) on a
reduced-width model over DummyDataset samples (vla/datasets.py, reference datasets.py:177-232): the loss must fall when
over-fitting a handful of samples, the metrics file and the native-format checkpoint must be written."""
import json

import numpy as np
import pytest
import torch

from test_data_cpu import WordTokenizer

pytestmark = pytest.mark.gpu


class Repeat(torch.utils.data.IterableDataset):
    def __init__(self, ds, n_samples, n_batches, batch):
        self.ds, self.n_samples, self.total = ds, n_samples, n_batches * batch

    def __len__(self):
        return self.total

    def __iter__(self):
        for i in range(self.total):
            s = dict(self.ds[i % self.n_samples])
            s["dataset_name"] = b"dummy_a" if (i % 2) else b"dummy_b"
            yield s


@pytest.mark.parametrize("train_strategy", ["fsdp-shard-grad-op", "fsdp-full-shard"])
def test_run_vla_training_overfits_and_checkpoints(dev, tmp_path, train_strategy):
    """`fsdp-full-shard`: the decoder layers' parameters are sharded (gathered per layer around their use, the model's
    own layer allocation freed during training, gathered back by `finish()`); everything observable stays the same."""
    from bridgelang_amd import weights as W
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import OpenVLAForActionPrediction
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor
    from bridgelang_amd.training.checkpoint import from_model_state_dicts
    from bridgelang_amd.training.strategy import VLAMetrics, get_train_strategy, lr_at
    from bridgelang_amd.util.data_utils import PaddedCollatorForActionPrediction
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    from bridgelang_amd.vla.datasets import DummyDataset
    dims = W.tiny_dims()
    vlm = OpenVLAForActionPrediction(OpenVLAConfig(norm_stats={}), device=dev, dims=dims).init_synthetic(seed=1)
    tok = WordTokenizer()
    at = ActionTokenizer(tok)
    ds = DummyDataset(at, tok, PrismaticImageProcessor().apply_transform, length=64, seed=0)
    B, steps = 4, 12
    data = Repeat(ds, n_samples=4, n_batches=steps, batch=B)
    collator = PaddedCollatorForActionPrediction(2048, tok.pad_token_id, padding_side="right")
    strat = get_train_strategy(train_strategy, vlm=vlm, device_id=0, stage="vla-train", epochs=1, max_steps=steps,
                               global_batch_size=B, per_device_batch_size=B, learning_rate=1e-3, weight_decay=0.0,
                               max_grad_norm=1.0, lr_scheduler_type="constant", warmup_ratio=0.0, max_text_len=8)
    before = {k: v.float().cpu() for k, v in vlm.state_dict().items()}
    strat.run_setup(tmp_path, n_train_examples=len(data))
    assert vlm.weights.layers_resident == (train_strategy != "fsdp-full-shard")
    metrics = VLAMetrics(("jsonl",), "run0", tmp_path, {"lr": 1e-3})
    strat.run_vla_training(data, collator, at, metrics, save_interval=1000, save_full_model=False)
    assert vlm.weights.layers_resident                      # finish() brought a sharded model back
    rows = [json.loads(l) for l in open(tmp_path / "run0.jsonl")]
    losses = [r["VLA Train/Loss"] for r in rows]
    print("losses", [round(x, 3) for x in losses], "acc", [round(r["VLA Train/Action Token Accuracy"], 2) for r in rows])
    assert len(rows) == steps and rows[-1]["VLA Train/Step"] == steps
    assert losses[-1] < 0.5 * losses[0], losses
    assert rows[-1]["VLA Train/Action Token Accuracy"] >= rows[0]["VLA Train/Action Token Accuracy"]
    assert all(np.isfinite(r["VLA Train/L1 Loss"]) and r["VLA Train/Step Time"] > 0 for r in rows)
    assert "dummy_a/L1 Loss" in rows[-1] and "dummy_b/Action Token Accuracy" in rows[-1]
    assert strat.step_engine.L >= 16                       # re-planned past max_text_len=8 on the first batch
    # checkpoint: reference-native module / key names, trainable modules only, fp32 masters
    ckpts = list((tmp_path / "checkpoints").glob("step-000012-epoch-*-loss=*.pt"))
    assert len(ckpts) == 1, list((tmp_path / "checkpoints").iterdir())
    ck = torch.load(ckpts[0], weights_only=True)["model"]
    assert set(ck) == {"projector", "llm_backbone"}
    assert "projector.0.weight" in ck["projector"] and "llm.model.layers.0.self_attn.q_proj.weight" in ck["llm_backbone"]
    hf = from_model_state_dicts(ck)
    live = vlm.state_dict()
    moved = 0
    for k, v in hf.items():
        assert v.dtype == torch.float32 and torch.equal(v.to(torch.bfloat16).float(), live[k].float().cpu()), k
        moved += int((v - before[k]).abs().max() > 0)
    assert moved == len(hf)
    frozen = [k for k in before if k.startswith("vision_backbone.")]
    assert all(torch.equal(before[k], live[k].float().cpu()) for k in frozen)
    # schedules
    assert lr_at(0, 1.0, "linear-warmup+cosine-decay", 100, 10) == 0.0 and lr_at(10, 1.0, "linear-warmup+cosine-decay", 100, 10) == 1.0
    assert abs(lr_at(55, 1.0, "linear-warmup+cosine-decay", 100, 10) - 0.5) < 1e-9


def test_lora_finetune_loop(dev, tmp_path):
    """finetune.py loop: LoRA adapters over DummyDataset batches; loss falls, adapter + merged checkpoints load back."""
    from safetensors.torch import load_file
    from bridgelang_amd import weights as W
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import OpenVLAForActionPrediction
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor
    from bridgelang_amd.training.finetune import FinetuneConfig, finetune
    from bridgelang_amd.util.data_utils import PaddedCollatorForActionPrediction
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    from bridgelang_amd.vla.datasets import DummyDataset
    dims = W.tiny_dims()
    vlm = OpenVLAForActionPrediction(OpenVLAConfig(norm_stats={}), device=dev, dims=dims).init_synthetic(seed=1)
    base = {k: v.float().cpu() for k, v in vlm.state_dict().items()}
    tok = WordTokenizer()
    at = ActionTokenizer(tok)
    ds = DummyDataset(at, tok, PrismaticImageProcessor().apply_transform, length=64, seed=0)
    B, steps = 4, 10
    loader = torch.utils.data.DataLoader(Repeat(ds, 4, steps, B), batch_size=B,
                                         collate_fn=PaddedCollatorForActionPrediction(2048, tok.pad_token_id, padding_side="right"))
    cfg = FinetuneConfig(run_root_dir=tmp_path / "run", adapter_tmp_dir=tmp_path / "adapter", batch_size=B, max_steps=steps,
                         learning_rate=2e-3, log_every=1)
    out = finetune(vlm, loader, at, cfg, log_path=tmp_path / "log.jsonl")
    rows = [json.loads(l) for l in open(tmp_path / "log.jsonl")]
    print("lora losses", [round(r["train_loss"], 3) for r in rows])
    assert out["steps"] == steps and rows[-1]["train_loss"] < 0.8 * rows[0]["train_loss"]
    ad = load_file(out["adapter"])
    assert "base_model.model.language_model.model.layers.0.self_attn.q_proj.lora_B.weight" in ad
    assert any(v.abs().sum() > 0 for k, v in ad.items() if "lora_B" in k)
    merged = load_file(out["model"])
    assert set(merged) == set(base)
    k = "language_model.model.layers.0.self_attn.q_proj.weight"
    assert not torch.equal(merged[k].float(), base[k])                        # adapters merged in
    assert torch.equal(merged["language_model.lm_head.weight"].float(), base["language_model.lm_head.weight"])
    assert all(torch.equal(vlm.state_dict()[n].float().cpu(), base[n]) for n in (k, "projector.fc1.weight"))   # base frozen
