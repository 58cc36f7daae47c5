#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING the reference files that import in this container:
action tokenizer, padded collator, prompt builder, fused-MLP projector, OpenVLAConfig. Files are loaded by path (the
package `prismatic/__init__.py` pulls in draccus, which is absent). Run here only — /root/reference does not exist on the
GPU box; the fixtures (data: inputs + expected outputs) are what travels.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.util
import json
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def load(rel: str, name: str):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class StubTokenizer:
    """The only tokenizer surface ActionTokenizer touches: vocab_size, decode, batch_decode."""
    vocab_size = 32000

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)

    def batch_decode(self, rows):
        return [self.decode(r) for r in rows]


def main() -> None:
    # ---- ActionTokenizer (prismatic/vla/action_tokenizer.py) ----
    at = load("prismatic/vla/action_tokenizer.py", "ref_action_tokenizer").ActionTokenizer(StubTokenizer())
    grid = np.concatenate([np.linspace(-1.2, 1.2, 4801), at.bins, at.bin_centers,
                           np.nextafter(at.bins, np.inf), np.nextafter(at.bins, -np.inf)])
    enc = np.array([int(x) for x in at(grid).split(" ")], dtype=np.int64)
    ids = np.arange(31744 - 300, 32064, dtype=np.int64)
    dec = at.decode_token_ids_to_actions(ids)
    batch = np.stack([grid[:7], grid[100:107], grid[-7:]])
    enc_batch = np.array([[int(x) for x in row.split(" ")] for row in at(batch)], dtype=np.int64)
    np.savez(OUT / "action_tokenizer.npz", grid=grid, enc=enc, ids=ids, dec=dec, bins=at.bins, centers=at.bin_centers,
             begin_idx=np.int64(at.action_token_begin_idx), batch=batch, enc_batch=enc_batch)

    # ---- PaddedCollatorForActionPrediction (prismatic/util/data_utils.py) ----
    du = load("prismatic/util/data_utils.py", "ref_data_utils")
    coll = du.PaddedCollatorForActionPrediction(model_max_length=12, pad_token_id=32000)
    g = torch.Generator().manual_seed(0)
    lens = [5, 12, 15, 1]
    inst = []
    for i, n in enumerate(lens):
        ids_ = torch.randint(1, 31999, (n,), generator=g)
        lab = ids_.clone()
        lab[: max(n - 3, 0)] = -100
        inst.append(dict(input_ids=ids_, labels=lab, pixel_values=torch.full((6, 2, 2), float(i)), dataset_name=f"d{i}"))
    out = coll(inst)
    np.savez(OUT / "collator.npz", lens=np.array(lens), **{f"in_ids_{i}": x["input_ids"].numpy() for i, x in enumerate(inst)},
             **{f"in_lab_{i}": x["labels"].numpy() for i, x in enumerate(inst)}, input_ids=out["input_ids"].numpy(),
             labels=out["labels"].numpy(), attention_mask=out["attention_mask"].numpy(),
             pixel_values=out["pixel_values"].numpy())

    # ---- PurePromptBuilder (prismatic/models/backbones/llm/prompting/base_prompter.py) ----
    bp = load("prismatic/models/backbones/llm/prompting/base_prompter.py", "ref_base_prompter")
    prompts = {}
    for instr in ["grasp the snack bag", "Put The Carrot On The Plate", "  do something spectacular  ", ""]:
        b = bp.PurePromptBuilder("openvla")
        t1 = b.add_turn("human", f"What action should the robot take to {instr.lower()}?")
        p_after_human = b.get_prompt()
        t2 = b.add_turn("gpt", "<act>")
        prompts[instr] = dict(turn1=t1, prompt1=p_after_human, turn2=t2, prompt2=b.get_prompt(),
                              potential=bp.PurePromptBuilder("openvla").get_potential_prompt("hello <image> there"))
    (OUT / "prompts.json").write_text(json.dumps(prompts, indent=1))

    # ---- FusedMLPProjector (prismatic/util/nn_utils.py) forward, fp32 ----
    nn_utils = load("prismatic/util/nn_utils.py", "ref_nn_utils")
    torch.manual_seed(2176)
    proj = nn_utils.FusedMLPProjector(fused_vision_dim=96, llm_dim=64)
    x = torch.randn(2, 5, 96, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = proj(x)
    sd = {k: v.numpy() for k, v in proj.state_dict().items()}
    np.savez(OUT / "projector.npz", x=x.numpy(), y=y.numpy(), **{k.replace(".", "__"): v for k, v in sd.items()})

    # ---- OpenVLAConfig (prismatic/extern/hf/configuration_prismatic.py) ----
    cp = load("prismatic/extern/hf/configuration_prismatic.py", "ref_configuration_prismatic")
    stats = {"bridge_orig": {"action": {"q01": [-0.1] * 7, "q99": [0.2] * 7, "mask": [True] * 6 + [False]}}}
    cfg = cp.OpenVLAConfig(vision_backbone_id="dinosiglip-vit-so-224px", llm_backbone_id="llama2-7b-pure",
                           arch_specifier="no-align+fused-gelu-mlp", image_resize_strategy="resize-naive",
                           text_config=dict(vocab_size=32064, pad_token_id=32000), norm_stats=stats)
    tc = cfg.text_config
    fields = dict(model_type=cfg.model_type, use_fused_vision_backbone=cfg.use_fused_vision_backbone,
                  timm_model_ids=cfg.timm_model_ids, timm_override_act_layers=cfg.timm_override_act_layers,
                  image_sizes=cfg.image_sizes, hf_llm_id=cfg.hf_llm_id, llm_max_length=cfg.llm_max_length,
                  pad_token_id=cfg.pad_token_id, pad_to_multiple_of=cfg.pad_to_multiple_of,
                  n_action_bins=cfg.n_action_bins, norm_stats=cfg.norm_stats,
                  text=dict(vocab_size=tc.vocab_size, hidden_size=tc.hidden_size, intermediate_size=tc.intermediate_size,
                            num_hidden_layers=tc.num_hidden_layers, num_attention_heads=tc.num_attention_heads,
                            rms_norm_eps=tc.rms_norm_eps, max_position_embeddings=tc.max_position_embeddings,
                            hidden_act=tc.hidden_act))
    (OUT / "openvla_config.json").write_text(json.dumps(fields, indent=1))

    # ---- predict_action tail (modeling_prismatic.py:521-534 cannot be imported: timm) — evaluated here with the
    #      reference's ActionTokenizer decode + the formula's inputs, so the fixture pins bin centres and un-normalisation
    tok_ids = np.array([31744, 31745, 31872, 31999, 32000, 31800, 31900], dtype=np.int64)
    centers = at.decode_token_ids_to_actions(tok_ids)
    q01, q99 = np.array([-0.1, -0.2, -0.3, -1, -1, -1, 0.0]), np.array([0.2, 0.3, 0.4, 1, 1, 1, 1.0])
    mask = np.array([True] * 6 + [False])
    actions = np.where(mask, 0.5 * (centers + 1) * (q99 - q01) + q01, centers)
    np.savez(OUT / "unnormalize.npz", tok_ids=tok_ids, centers=centers, q01=q01, q99=q99, mask=mask, actions=actions)
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")), sorted(p.name for p in OUT.glob("*.json")))


if __name__ == "__main__":
    main()
