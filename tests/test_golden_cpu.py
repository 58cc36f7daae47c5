"""CPU: pin the host-side classes and the oracle's INT/f64 action head against fixtures captured from the reference
itself (tests/golden/make_golden.py imports the reference files that import offline)."""
import json
from pathlib import Path

import numpy as np
import torch

from oracle import restate as R

G = Path(__file__).resolve().parent / "golden"


class StubTokenizer:
    vocab_size = 32000

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)

    def batch_decode(self, rows):
        return [self.decode(r) for r in rows]


def test_action_tokenizer_matches_reference():
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    z = np.load(G / "action_tokenizer.npz")
    at = ActionTokenizer(StubTokenizer())
    assert np.array_equal(at.bins, z["bins"]) and np.array_equal(at.bin_centers, z["centers"])
    assert at.action_token_begin_idx == int(z["begin_idx"]) == 31743 and at.vocab_size == 256
    assert np.array_equal(at.encode_ids(z["grid"]), z["enc"])                       # bit-exact ids
    assert at(z["grid"]) == " ".join(str(i) for i in z["enc"])                      # string path
    assert at(z["batch"]) == [" ".join(str(i) for i in row) for row in z["enc_batch"]]
    assert np.array_equal(at.decode_token_ids_to_actions(z["ids"]), z["dec"])       # f64 exact
    # the oracle's restatement of the same arithmetic
    assert np.array_equal(R.encode_actions(z["grid"], 32000), z["enc"])
    # SURVEY App. A.4 known answers
    assert list(at.encode_ids(np.array([-1, -0.999, -0.5, 0, 0.5, 0.9999, 1]))) == [31999, 31999, 31936, 31872, 31808, 31745, 31744]


def test_unnormalize_matches_reference():
    z = np.load(G / "unnormalize.npz")
    stats = {"q01": z["q01"], "q99": z["q99"], "mask": z["mask"]}
    got = R.decode_actions(z["tok_ids"], 32000, stats)
    assert np.array_equal(got, z["actions"])
    stats_nomask = {"q01": z["q01"], "q99": z["q99"]}
    got2 = R.decode_actions(z["tok_ids"], 32000, stats_nomask)
    assert np.array_equal(got2, 0.5 * (z["centers"] + 1) * (z["q99"] - z["q01"]) + z["q01"])


def test_collator_matches_reference():
    from bridgelang_amd.util.data_utils import PaddedCollatorForActionPrediction
    z = np.load(G / "collator.npz")
    inst = [dict(input_ids=torch.from_numpy(z[f"in_ids_{i}"]), labels=torch.from_numpy(z[f"in_lab_{i}"]),
                 pixel_values=torch.full((6, 2, 2), float(i)), dataset_name=f"d{i}") for i in range(len(z["lens"]))]
    out = PaddedCollatorForActionPrediction(model_max_length=12, pad_token_id=32000)(inst)
    assert np.array_equal(out["input_ids"].numpy(), z["input_ids"])
    assert np.array_equal(out["labels"].numpy(), z["labels"])
    assert np.array_equal(out["attention_mask"].numpy(), z["attention_mask"])
    assert np.array_equal(out["pixel_values"].numpy(), z["pixel_values"])
    assert out["dataset_names"] == ["d0", "d1", "d2", "d3"]
    # dict-valued pixel_values (native path) are stacked per key
    inst2 = [dict(input_ids=x["input_ids"], labels=x["labels"], pixel_values={"dino": x["pixel_values"][:3], "siglip": x["pixel_values"][3:]}) for x in inst]
    out2 = PaddedCollatorForActionPrediction(12, 32000)(inst2)
    assert out2["pixel_values"]["dino"].shape == (4, 3, 2, 2) and "dataset_names" not in out2


def test_prompt_builder_matches_reference():
    from bridgelang_amd.models.prompting import PurePromptBuilder, vla_prompt
    ref = json.loads((G / "prompts.json").read_text())
    for instr, exp in ref.items():
        b = PurePromptBuilder("openvla")
        assert b.add_turn("human", f"What action should the robot take to {instr.lower()}?") == exp["turn1"]
        assert b.get_prompt() == exp["prompt1"] == vla_prompt(instr)
        assert b.add_turn("gpt", "<act>") == exp["turn2"]
        assert b.get_prompt() == exp["prompt2"]
        assert PurePromptBuilder("openvla").get_potential_prompt("hello <image> there") == exp["potential"]


def test_projector_oracle_matches_reference():
    """oracle.projector in fp32 mode reproduces the reference FusedMLPProjector forward."""
    z = np.load(G / "projector.npz")
    sd = {"projector.fc1.weight": z["projector__0__weight"], "projector.fc1.bias": z["projector__0__bias"],
          "projector.fc2.weight": z["projector__2__weight"], "projector.fc2.bias": z["projector__2__bias"],
          "projector.fc3.weight": z["projector__4__weight"], "projector.fc3.bias": z["projector__4__bias"]}
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    y = R.projector(R.Prec(False), sd, torch.from_numpy(z["x"]))
    assert torch.allclose(y, torch.from_numpy(z["y"]), rtol=1e-5, atol=1e-6)


def test_config_matches_reference():
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import dims_from_config
    ref = json.loads((G / "openvla_config.json").read_text())
    cfg = OpenVLAConfig(norm_stats=ref["norm_stats"], text_config=dict(vocab_size=32064, pad_token_id=32000))
    for k in ("model_type", "use_fused_vision_backbone", "timm_model_ids", "timm_override_act_layers", "image_sizes",
              "hf_llm_id", "llm_max_length", "pad_token_id", "pad_to_multiple_of", "n_action_bins", "norm_stats"):
        assert getattr(cfg, k) == ref[k], k
    tc = cfg.text_config
    for k, v in ref["text"].items():
        assert getattr(tc, k) == v, k
    d = dims_from_config(cfg)
    assert (d.llm_dim, d.llm_layers, d.llm_heads, d.llm_inter, d.vocab) == (4096, 32, 32, 11008, 32064)
    assert d.rms_eps == ref["text"]["rms_norm_eps"] and d.max_pos == 2048
    # default-constructed config carries the same text model as the reference's converted checkpoint
    d2 = dims_from_config(OpenVLAConfig(norm_stats=ref["norm_stats"]))
    assert d2 == d
    import pytest
    with pytest.raises(ValueError):
        OpenVLAConfig(vision_backbone_id="not-a-backbone")
