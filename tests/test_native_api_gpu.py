"""The reference's native surface on the HIP path: registries → OpenVLA(image, instruction) ≡ the HF-class path on the
same inputs; both on-disk layouts (native run dir `.pt`, HF safetensors incl. sharded index) round-trip the weights."""
import json

import numpy as np
import pytest
import torch
from PIL import Image

from test_data_cpu import WordTokenizer

pytestmark = pytest.mark.gpu
STATS = {"bridge_orig": {"action": {"q01": [-0.5] * 7, "q99": [0.7] * 7, "mask": [True] * 6 + [False]}}}


def build(dev, seed=4):
    from bridgelang_amd import weights as W
    from bridgelang_amd.models.materialize import get_llm_backbone_and_tokenizer, get_vision_backbone_and_transform
    from bridgelang_amd.models.vlms import OpenVLA
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    tok = WordTokenizer()
    vb, _ = get_vision_backbone_and_transform("dinosiglip-vit-so-224px", "resize-naive")
    lb, _ = get_llm_backbone_and_tokenizer("llama2-7b-pure", tokenizer=tok)
    vla = OpenVLA("openvla-tiny", vb, lb, norm_stats=STATS, action_tokenizer=ActionTokenizer(tok), device=dev, dims=W.tiny_dims())
    vla.hf.init_synthetic(seed)
    return vla, tok


def test_native_predict_action_matches_hf_path(dev):
    vla, tok = build(dev)
    img = Image.fromarray(np.random.RandomState(0).randint(0, 256, (224, 224, 3), dtype=np.uint8))
    act = vla.predict_action(img, "Grasp the snack bag", unnorm_key="bridge_orig")
    assert act.shape == (7,) and np.isfinite(act).all()
    prompt = vla.get_prompt_builder()
    prompt.add_turn(role="human", message="What action should the robot take to grasp the snack bag?")
    ids = tok(prompt.get_prompt(), return_tensors="pt")["input_ids"]
    pv = vla.vision_backbone.image_transform(img)
    pv6 = torch.cat([pv["dino"], pv["siglip"]])[None].to(torch.bfloat16)
    ref = vla.hf.predict_action(ids.to(dev), unnorm_key="bridge_orig", pixel_values=pv6.to(dev))
    assert np.array_equal(act, ref)
    feats = vla.vision_backbone(pv6)                               # backbone plugin forward, bound to the VLM
    assert feats.shape == (1, 256, vla.dims.vision_dim) and feats.dtype == torch.bfloat16
    out = vla(input_ids=ids.to(dev), pixel_values={k: v[None] for k, v in pv.items()}, labels=ids.to(dev))
    assert out.logits.shape[:2] == (1, 256 + ids.shape[1]) and torch.isfinite(out.loss)
    vla.freeze_backbones("vla-full-train")
    assert vla.trainable_module_keys == ["vision_backbone", "projector", "llm_backbone"] and vla.vision_backbone_requires_grad
    with pytest.raises(ValueError):
        vla.freeze_backbones("nonsense")


def test_native_and_hf_layouts_round_trip(dev, tmp_path):
    from bridgelang_amd.models.load import from_pretrained, load_vla, save_pretrained, save_run_dir
    vla, tok = build(dev, seed=9)
    ref = {k: v.cpu() for k, v in vla.hf.state_dict().items()}
    ck = save_run_dir(vla, tmp_path / "run", step=12, epoch=1, loss=0.25)
    assert ck.name == "step-000012-epoch-01-loss=0.2500.pt"
    raw = torch.load(ck, weights_only=True)["model"]
    assert set(raw) == {"vision_backbone", "llm_backbone", "projector"} and "projector.4.bias" in raw["projector"]
    assert any(k.endswith("ls1.gamma") for k in raw["vision_backbone"]) and "llm.model.norm.weight" in raw["llm_backbone"]
    back = load_vla(tmp_path / "run", tokenizer=tok, device=dev, dims=vla.dims)
    assert back.norm_stats == json.loads(json.dumps(STATS))
    got = back.hf.state_dict()
    assert all(torch.equal(got[k].cpu(), ref[k]) for k in ref)
    # HF export, forced into several shards
    save_pretrained(vla.hf, tmp_path / "hf", max_shard_bytes=8 << 20)
    assert (tmp_path / "hf" / "model.safetensors.index.json").exists()
    m = from_pretrained(tmp_path / "hf", device=dev, dims=vla.dims)
    got = m.state_dict()
    assert all(torch.equal(got[k].cpu(), ref[k]) for k in ref) and m.norm_stats == STATS
    with pytest.raises((FileNotFoundError, ValueError)):
        from_pretrained("openvla/openvla-7b", device=dev)
