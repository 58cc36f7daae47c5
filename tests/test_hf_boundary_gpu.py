"""The HF drop-in boundary on the GPU: the reference demo's call sequence
(run_openvla_demo.py:21-44) against a `save_pretrained` directory, the PreTrainedModel plumbing over the weight arena,
batched generation with right-padded prompts, bounded engine caches, and the entry scripts end to end on DummyDataset."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
STATS = {"bridge_orig": {"action": {"q01": [-1.0, -0.5, -0.25, 0.0, -1.0, -2.0, 0.0], "q99": [1.0, 0.5, 0.75, 2.0, 1.0, 2.0, 1.0],
                                   "mask": [True] * 6 + [False]}}}
SMALL_LLM = dict(vocab_size=32064, pad_token_id=32000, rms_norm_eps=1e-6, hidden_size=512, intermediate_size=1536,
                 num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4)


@pytest.fixture(scope="module")
def saved(dev, tmp_path_factory):
    """Full-geometry towers (timm ids fix them) + a 2-layer / 512-wide Llama, synthetic weights, written to disk in the
    HF layout with its processor files."""
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import OpenVLAForActionPrediction
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor, PrismaticProcessor
    from bridgelang_amd.util.synthetic_tokenizer import SyntheticLlamaTokenizer
    d = tmp_path_factory.mktemp("openvla-small")
    model = OpenVLAForActionPrediction(OpenVLAConfig(norm_stats=STATS, text_config=SMALL_LLM), device=dev).init_synthetic(seed=5)
    model.save_pretrained(d, max_shard_size="400MB")
    PrismaticProcessor(PrismaticImageProcessor(), SyntheticLlamaTokenizer()).save_pretrained(d)
    return d, model


def test_reference_demo_call_sequence(saved, dev):
    """run_openvla_demo.py:21-44 with only the import / path lines changed."""
    import transformers
    from PIL import Image
    from prismatic.extern.hf.modeling_prismatic import register_auto_classes
    from prismatic.extern.hf.processing_prismatic import PrismaticProcessor
    model_path, original = saved
    assert len(list(Path(model_path).glob("model-*.safetensors"))) > 1 and (Path(model_path) / "model.safetensors.index.json").exists()
    register_auto_classes()
    Auto = getattr(transformers, "AutoModelForVision2Seq", None) or transformers.AutoModelForImageTextToText
    processor = PrismaticProcessor.from_pretrained(model_path, trust_remote_code=True)
    vla = Auto.from_pretrained(model_path, trust_remote_code=True, attn_implementation="sdpa", torch_dtype=torch.bfloat16,
                               low_cpu_mem_usage=True).to("cuda:0").eval()
    image = Image.fromarray(np.random.RandomState(0).randint(0, 256, (561, 772, 3), dtype=np.uint8)).convert("RGB")
    prompt = "In: What action should the robot take to grasp the snack bag?\nOut:"
    with torch.no_grad():
        inputs = processor(prompt, image).to("cuda:0", dtype=torch.bfloat16)
        action = vla.predict_action(**inputs, unnorm_key="bridge_orig", do_sample=False, use_cache=False)
    assert isinstance(action, np.ndarray) and action.shape == (7,) and action.dtype == np.float64
    want = original.predict_action(**inputs, unnorm_key="bridge_orig", do_sample=False)
    assert np.array_equal(action, want), "the reloaded checkpoint must reproduce the saved model's action"
    lo, hi = np.array(STATS["bridge_orig"]["action"]["q01"]), np.array(STATS["bridge_orig"]["action"]["q99"])
    assert ((action[:6] >= lo[:6] - 1e-9) & (action[:6] <= hi[:6] + 1e-9)).all() and -1.0 <= action[6] <= 1.0
    # HF plumbing over the arena
    assert isinstance(vla, transformers.PreTrainedModel) and vla.device == torch.device("cuda:0") and vla.dtype == torch.bfloat16
    assert vla.training is False and list(vla.parameters()) == [] and vla.config.image_sizes == [224, 224]
    assert vla.vision_backbone.featurizer.patch_embed.num_patches == 256
    with pytest.raises(NotImplementedError):
        vla.to(torch.float32)
    sd = vla.state_dict()
    assert "vision_backbone.fused_featurizer.attn_pool.kv.weight" in sd and "vision_backbone.featurizer.blocks.23.ls2.scale_factor" in sd
    a, b = sd["language_model.lm_head.weight"], original.state_dict()["language_model.lm_head.weight"]
    assert torch.equal(a, b)
    # forward() returns fresh tensors: a second call must not overwrite the first call's logits
    o1 = vla(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"], pixel_values=inputs["pixel_values"])
    keep = o1.logits.clone()
    vla(input_ids=inputs["input_ids"].flip(1), attention_mask=inputs["attention_mask"], pixel_values=inputs["pixel_values"])
    assert torch.equal(o1.logits, keep)


def test_generate_with_right_padded_prompts_equals_unpadded(saved, dev):
    """A coalesced batch of prompts of different lengths (right-padded, attention_mask) gives every sequence exactly the
    ids and logits it gets alone (the reference's cached branch is batch-1 only, modeling_prismatic.py:326,460-463)."""
    _, model = saved
    g = torch.Generator().manual_seed(3)
    lens, L = [12, 7, 9, 12], 12
    pv = (torch.rand(4, 6, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    ids = torch.randint(3, 31743, (4, L), generator=g)
    ids[:, 0] = 1
    mask = torch.zeros(4, L, dtype=torch.long)
    for b, n in enumerate(lens):
        ids[b, n - 1] = 29871
        ids[b, n:] = 32000
        mask[b, :n] = 1
    out = model.generate(ids.to(dev), max_new_tokens=7, pixel_values=pv, attention_mask=mask.to(dev))
    assert tuple(out.shape) == (4, L + 7)
    padded_logits = model.engine(4, L, 7, padded=True).logits.clone()        # [7, 4, V]
    for b, n in enumerate(lens):
        alone = model.generate(ids[b:b + 1, :n].to(dev), max_new_tokens=7, pixel_values=pv[b:b + 1])
        assert torch.equal(alone[0, n:], out[b, L:]), f"sequence {b} (length {n}): ids differ from the un-padded run"
        assert torch.equal(model.engine(1, n, 7).logits[:, 0], padded_logits[:, b]), f"sequence {b}: logits differ"
    with pytest.raises(ValueError, match="right-padded"):
        bad = mask.clone(); bad[1, 0] = 0
        model.generate(ids.to(dev), max_new_tokens=7, pixel_values=pv, attention_mask=bad.to(dev))
    # predict_action on the padded batch: [B, 7] actions, the batch-1 call's row
    acts = model.predict_action(input_ids=ids.to(dev), pixel_values=pv, attention_mask=mask.to(dev), unnorm_key="bridge_orig")
    one = model.predict_action(input_ids=ids[1:2, :lens[1]].to(dev), pixel_values=pv[1:2], unnorm_key="bridge_orig")
    assert acts.shape == (4, 7) and np.array_equal(acts[1], one)


def test_engine_caches_are_bounded(saved, dev):
    """Many distinct (batch, prompt length) pairs — what a server sees — must not grow HBM use without bound."""
    _, model = saved
    g = torch.Generator().manual_seed(0)
    pv = (torch.rand(2, 6, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    peak = []
    for L in range(6, 30):
        ids = torch.randint(3, 31743, (2, L), generator=g)
        ids[:, 0], ids[:, -1] = 1, 29871
        model.generate(ids.to(dev), max_new_tokens=7, pixel_values=pv)
        model(input_ids=ids.to(dev), pixel_values=pv)
        torch.cuda.synchronize()
        peak.append(torch.cuda.memory_allocated())
    assert len(model._engines) <= model._ENGINE_CACHE and len(model._forward_engines) <= model._ENGINE_CACHE
    assert max(peak[12:]) <= 1.15 * max(peak[:12]), "allocation keeps growing with new shapes"


@pytest.mark.parametrize("script", ["finetune", "train"])
def test_entry_scripts_run_two_steps_on_dummy_data(script, tmp_path):
    """vla-scripts/{finetune,train}.py with the reference's flags, synthetic tiny checkpoint, DummyDataset, 2-3 steps."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29588")
    if script == "finetune":
        cmd = ["--vla_path", "synthetic:openvla-tiny", "--dataset_name", "dummy", "--run_root_dir", str(tmp_path / "runs"),
               "--adapter_tmp_dir", str(tmp_path / "adapters"), "--lora_rank", "32", "--batch_size", "4", "--max_steps", "3",
               "--save_steps", "2", "--learning_rate", "5e-4", "--image_aug", "False", "--dummy_length", "16"]
    else:
        cmd = ["--vla.type", "prism-dinosiglip-224px+mx-bridge", "--vla.expected_world_size", "1", "--vla.global_batch_size", "4",
               "--vla.per_device_batch_size", "4", "--vla.max_steps", "2", "--vla.data_mix", "dummy", "--vla.base_vlm", "openvla-tiny",
               "--synthetic_init", "True", "--run_root_dir", str(tmp_path / "runs"), "--run_id", "smoke", "--save_interval", "2",
               "--dummy_length", "16", "--is_resume", "False"]
    p = subprocess.run([sys.executable, str(ROOT / "vla-scripts" / f"{script}.py")] + cmd, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    if script == "finetune":
        run = next((tmp_path / "runs").iterdir())
        rows = [json.loads(l) for l in open(run / "train_log.jsonl")]
        assert rows and all(np.isfinite(r["train_loss"]) for r in rows) and {"train_loss", "action_accuracy", "l1_loss"} <= set(rows[0])
        assert (run / "model.safetensors").exists() and (run / "dataset_statistics.json").exists()
        assert any((tmp_path / "adapters").rglob("adapter_model.safetensors"))
    else:
        run = tmp_path / "runs" / "smoke"
        cfgj = json.loads((run / "config.json").read_text())
        assert cfgj["vla"]["type"] == "prism-dinosiglip-224px+mx-bridge" and (run / "config.yaml").exists()
        rows = [json.loads(l) for l in open(run / "smoke.jsonl")]
        assert rows[-1]["VLA Train/Step"] == 2 and np.isfinite(rows[-1]["VLA Train/Loss"])
        assert list((run / "checkpoints").glob("step-000002-epoch-*-loss=*.pt")) and (run / "dataset_statistics.json").exists()


def _small_inputs(dev, B=3, L=12, seed=11):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 31743, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 1, 29871
    pv = (torch.rand(B, 6, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16)
    return ids.to(dev), pv.to(dev)


def test_forward_step_by_step_reproduces_generate(saved, dev):
    """The HF cached-forward surface (modeling_prismatic.py:325-341, 450-485): a caller that drives
    forward() itself — multimodal call with use_cache=True, then one [B, 1] call per token with the returned
    past_key_values, inputs built by prepare_inputs_for_generation — gets, bit for bit, the ids AND logits generate()
    computes on-device."""
    _, model = saved
    ids, pv = _small_inputs(dev)
    B, n = ids.shape[0], 7
    want = model.generate(ids, max_new_tokens=n, pixel_values=pv)[:, -n:].clone()
    eng = model.engine(B, ids.shape[1], n)
    want_logits = eng.logits.clone()                                  # [n, B, V]
    seq, past, got_logits = ids, None, []
    for t in range(n):
        kw = model.prepare_inputs_for_generation(seq, past_key_values=past, pixel_values=pv, attention_mask=None, use_cache=True)
        out = model(**kw)
        assert out.logits.shape == (B, 1, model.dims.vocab) and out.logits.dtype == torch.float32
        past = out.past_key_values
        assert past.get_seq_length() == eng.S + t
        got_logits.append(out.logits[:, -1])
        seq = torch.cat([seq, out.logits[:, -1].argmax(-1, keepdim=True)], dim=1)
    assert torch.equal(seq[:, -n:], want), "step-wise forward() must reproduce generate()'s ids"
    assert torch.equal(torch.stack(got_logits), want_logits), "… and its logits, bit for bit"
    # a token other than the model's own argmax can be fed (teacher forcing): the cache really is driven by the caller
    out0 = model(input_ids=ids, pixel_values=pv, use_cache=True)
    forced = torch.full((B, 1), 31900, dtype=torch.long, device=dev)
    o1 = model(input_ids=forced, past_key_values=out0.past_key_values, use_cache=True)
    ref = model.generate(torch.cat([ids, forced], 1), max_new_tokens=1, pixel_values=pv)[:, -1]
    # (prefill over L+1 tokens vs cached step: different kernels → compare ids only where the gap is decisive)
    top2 = o1.logits[:, -1].topk(2, -1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 0.05 * o1.logits.abs().max()
    assert torch.equal(o1.logits[:, -1].argmax(-1)[decisive], ref[decisive])
    # an action prediction of the SAME shape in between leaves the caller's cache alone (cached-forward engines have their own slots)
    held = model(input_ids=ids, pixel_values=pv, use_cache=True)
    model.generate(ids, max_new_tokens=model.cache_new_tokens, pixel_values=pv)
    again = model(input_ids=forced, past_key_values=held.past_key_values, use_cache=True)
    assert torch.equal(again.logits, o1.logits)
    # guards: capacity, stale handles, missing cache
    with pytest.raises(AssertionError):
        model(input_ids=forced, use_cache=True)
    stale = out0.past_key_values
    model(input_ids=ids, pixel_values=pv, use_cache=True)             # same engine slot prefilled again
    with pytest.raises(RuntimeError, match="stale"):
        model(input_ids=forced, past_key_values=stale)
    fresh = model(input_ids=ids, pixel_values=pv, use_cache=True).past_key_values
    for _ in range(model.cache_new_tokens - 1):
        model(input_ids=forced, past_key_values=fresh)
    with pytest.raises(RuntimeError, match="sized for"):
        model(input_ids=forced, past_key_values=fresh)


def test_forward_padded_prefill_then_cached_steps(saved, dev):
    """Right-padded batch through the step-wise surface ≡ generate() on the same padded batch."""
    _, model = saved
    ids, pv = _small_inputs(dev, B=2, L=14, seed=3)
    mask = torch.ones_like(ids)
    mask[1, 10:] = 0
    ids[1, 9] = 29871
    ids[1, 10:] = model.pad_token_id
    n = 5
    want = model.generate(ids, max_new_tokens=n, pixel_values=pv, attention_mask=mask)[:, -n:].clone()
    model.cache_new_tokens = n
    try:
        out = model(input_ids=ids, pixel_values=pv, attention_mask=mask, use_cache=True)
        toks = [out.logits[:, -1].argmax(-1)]
        for _ in range(n - 1):
            out = model(input_ids=toks[-1][:, None], past_key_values=out.past_key_values, use_cache=True)
            toks.append(out.logits[:, -1].argmax(-1))
    finally:
        model.cache_new_tokens = 7
    assert torch.equal(torch.stack(toks, 1), want)


def test_forward_language_only_inputs_embeds_and_hidden_states(saved, dev):
    """Language-only branch (pixel_values=None, modeling_prismatic.py:343-359) against the CPU oracle's Llama; the
    `inputs_embeds` form equals the `input_ids` form bit for bit; `output_hidden_states` returns HF's tuple."""
    from oracle import restate as R
    _, model = saved
    d = model.dims
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(3, 31743, (2, 9), generator=g)
    ids[:, 0] = 1
    labels = ids.clone()
    labels[:, :4] = -100
    out = model(input_ids=ids.to(dev), labels=labels.to(dev), output_hidden_states=True)
    assert out.logits.shape == (2, 9, d.vocab) and out.past_key_values is None and out.projector_features is None
    assert len(out.hidden_states) == d.llm_layers + 1 and out.hidden_states[0].shape == (2, 9, d.llm_dim)
    sd = {k: v.cpu() for k, v in model.state_dict().items() if k.startswith("language_model.")}
    p = R.Prec(True)
    x = sd["language_model.model.embed_tokens.weight"].float()[ids]
    assert torch.equal(out.hidden_states[0].float().cpu(), x)
    tr = []
    ref, _ = R.llama_forward(p, sd, x, d.llm_heads, d.llm_layers, d.rms_eps, d.rope_theta, trace=tr)
    scale = ref.abs().max()
    assert ((out.logits.cpu() - ref).abs().max() / scale) < 1.2e-2          # 2-layer bound of test_engine_gpu.py
    for l in range(d.llm_layers - 1):
        a, b = out.hidden_states[l + 1].float().cpu(), tr[l]
        assert ((a - b).abs().max() / b.abs().max()) < 1.2e-2
    last = R.rmsnorm(p, tr[-1], sd["language_model.model.norm.weight"].float(), d.rms_eps)
    assert ((out.hidden_states[-1].float().cpu() - last).abs().max() / last.abs().max()) < 1.2e-2
    # shifted CE (HF CausalLM loss) on the oracle's logits
    want_loss = torch.nn.functional.cross_entropy(ref[:, :-1].reshape(-1, d.vocab), labels[:, 1:].reshape(-1), ignore_index=-100)
    assert abs(float(out.loss) - float(want_loss)) < 2e-2
    emb = model.weights.embed[ids.to(dev)]
    out2 = model(inputs_embeds=emb)
    assert torch.equal(out2.logits, out.logits)
    with pytest.raises(NotImplementedError):
        model(input_ids=ids.to(dev), output_attentions=True)


@pytest.mark.parametrize("center_crop", [False, True])
def test_get_vla_action_device_path_equals_host_path(saved, dev, center_crop):
    """The robot loops' policy call (experiments/robot/openvla_utils.py:115-172) with the frame kept on the device — crop +
    resize, bicubic resize, dual normalisation all as HIP kernels — returns exactly the action of the host path that
    mirrors the reference (numpy crop → PIL → processor → predict_action), for a camera frame that is not 224 px."""
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor, PrismaticProcessor
    from bridgelang_amd.util.synthetic_tokenizer import SyntheticLlamaTokenizer
    from bridgelang_amd.vla.eval_preprocess import get_vla_action, vla_prompt
    _, model = saved
    processor = PrismaticProcessor(PrismaticImageProcessor(), SyntheticLlamaTokenizer())
    obs = {"full_image": np.random.RandomState(3).randint(0, 256, (256, 320, 3), dtype=np.uint8)}
    assert vla_prompt("openvla-7b", "Pick Up the Cup") == "In: What action should the robot take to pick up the cup?\nOut:"
    host = get_vla_action(model, processor, "openvla-7b", obs, "pick up the cup", "bridge_orig", center_crop=center_crop, on_device=False)
    devp = get_vla_action(model, processor, "openvla-7b", obs, "pick up the cup", "bridge_orig", center_crop=center_crop)
    assert host.shape == (7,) and np.array_equal(host, devp)
    # a frame that already lives in HBM is taken as is
    obs_dev = {"full_image": torch.from_numpy(obs["full_image"]).to(dev)}
    assert np.array_equal(get_vla_action(model, processor, "openvla-7b", obs_dev, "pick up the cup", "bridge_orig", center_crop=center_crop), host)
