"""Weight I/O in both on-disk layouts of the reference: the native run directory
(`config.json` + `checkpoints/*.pt` + `dataset_statistics.json`, prismatic/models/load.py:122-226) and the HF export
(`config.json` + `*.safetensors` [+ index], convert_openvla_weights_to_hf.py:235-249). Files are read with loaders that
execute nothing (`weights_only=True`, safetensors, json). Nothing is downloaded: ids that are not local paths raise."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Optional, Union

import torch

from ..extern.hf.configuration_prismatic import OpenVLAConfig
from ..extern.hf.modeling_prismatic import OpenVLAForActionPrediction
from ..vla.action_tokenizer import ActionTokenizer
from ..weights import VLADims
from .materialize import get_llm_backbone_and_tokenizer, get_vision_backbone_and_transform
from .vlms import OpenVLA


def _json_default(o):
    import numpy as np
    return o.tolist() if isinstance(o, np.ndarray) else str(o)


# ---- HF layout ---------------------------------------------------------------------------------------------------------
def save_pretrained(model: OpenVLAForActionPrediction, save_directory: Union[str, Path], max_shard_bytes: int = 5 << 30) -> None:
    from safetensors.torch import save_file
    d = Path(save_directory)
    d.mkdir(parents=True, exist_ok=True)
    c = model.config
    cfg = {"model_type": "openvla", "architectures": ["OpenVLAForActionPrediction"], "vision_backbone_id": c.vision_backbone_id,
           "llm_backbone_id": c.llm_backbone_id, "arch_specifier": c.arch_specifier,
           "use_fused_vision_backbone": c.use_fused_vision_backbone, "image_resize_strategy": c.image_resize_strategy,
           "llm_max_length": c.llm_max_length, "pad_token_id": c.pad_token_id, "pad_to_multiple_of": c.pad_to_multiple_of,
           "n_action_bins": c.n_action_bins, "norm_stats": c.norm_stats, "image_sizes": c.image_sizes,
           "timm_model_ids": c.timm_model_ids, "text_config": c.text_config.to_dict(), "torch_dtype": "bfloat16"}
    (d / "config.json").write_text(json.dumps(cfg, default=_json_default, indent=2))
    sd = {k: v.contiguous().cpu() for k, v in model.state_dict().items()}
    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        nb = v.numel() * v.element_size()
        if cur and size + nb > max_shard_bytes:
            shards.append(cur); cur, size = {}, 0
        cur[k] = v; size += nb
    shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], str(d / "model.safetensors"))
        return
    index = {"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": {}}
    for i, sh in enumerate(shards):
        name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, str(d / name))
        index["weight_map"].update({k: name for k in sh})
    (d / "model.safetensors.index.json").write_text(json.dumps(index, indent=2))


def load_hf_directory(cls, path: Union[str, Path], config=None, device: Union[str, torch.device, None] = None,
                      dims: Optional[VLADims] = None):
    """The body of `OpenVLAForActionPrediction.from_pretrained` (and of `AutoModelForVision2Seq.from_pretrained(local_dir,
    trust_remote_code=True)` once register_auto_classes() ran) for a LOCAL HF export."""
    from safetensors.torch import load_file
    d = Path(path)
    if not (d / "config.json").exists():
        raise FileNotFoundError(f"`{path}` is not a local model directory (nothing is fetched from the hub)")
    if config is None:
        raw = json.loads((d / "config.json").read_text())
        keep = ("vision_backbone_id", "llm_backbone_id", "arch_specifier", "use_fused_vision_backbone", "image_resize_strategy",
                "text_config", "llm_max_length", "pad_token_id", "pad_to_multiple_of", "norm_stats", "n_action_bins")
        tc = raw.get("text_config")
        if isinstance(tc, dict):
            raw["text_config"] = {k: v for k, v in tc.items() if k in (
                "vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                "num_key_value_heads", "rms_norm_eps", "rope_theta", "max_position_embeddings", "pad_token_id", "hidden_act")}
        kw = {k: raw[k] for k in keep if k in raw}
        if not issubclass(cls.config_class, OpenVLAConfig):
            kw.pop("norm_stats", None); kw.pop("n_action_bins", None)
        config = cls.config_class(**kw)
    model = cls(config, device=device, dims=dims)
    files = sorted(d.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under `{path}`")
    seen, carry = set(), {}
    for f in files:                                           # shard by shard: host memory stays near one shard
        sd = load_file(str(f))
        seen.update(sd)
        sd.update(carry)                                      # members of fused groups (q‖k‖v, gate/up) split across shards
        model.load_state_dict(sd, strict=False)
        carry = {}
        for g in model.weights.groups:
            have = [n for n in g.members if n in sd]
            if have and len(have) < len(g.members):
                carry.update({n: sd[n] for n in have})
    missing = [n for n in model.weights.placements if n not in seen]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
    return model.eval()


def from_pretrained(path: Union[str, Path], device: Union[str, torch.device, None] = None,
                    dims: Optional[VLADims] = None) -> OpenVLAForActionPrediction:
    """`AutoModelForVision2Seq.from_pretrained(local_dir, trust_remote_code=True)` for a LOCAL HF export."""
    return load_hf_directory(OpenVLAForActionPrediction, path, device=device, dims=dims)


# ---- native layout -----------------------------------------------------------------------------------------------------
def load_vla(model_id_or_path: Union[str, Path], hf_token: Optional[str] = None, cache_dir: Optional[Path] = None,
             load_for_training: bool = False, step_to_load: Optional[int] = None, model_type: str = "pretrained",
             tokenizer: Any = None, device: Union[str, torch.device] = "cuda:0", dims: Optional[VLADims] = None) -> OpenVLA:
    """prismatic/models/load.py:122-226 for a LOCAL run directory or checkpoint file."""
    p = Path(model_id_or_path)
    if p.is_file():
        assert p.suffix == ".pt" and p.parent.name == "checkpoints", "Invalid checkpoint!"
        run_dir, checkpoint_pt = p.parents[1], p
    elif p.is_dir():
        run_dir = p
        cands = sorted((run_dir / "checkpoints").glob("*.pt"))
        if step_to_load is not None:
            cands = [c for c in cands if c.name.startswith(f"step-{step_to_load:06d}")]
        latest = run_dir / "checkpoints" / "latest-checkpoint.pt"
        checkpoint_pt = latest if (latest.exists() and step_to_load is None) else (cands[-1] if cands else None)
        if checkpoint_pt is None:
            raise FileNotFoundError(f"no checkpoint under `{run_dir / 'checkpoints'}`")
    else:
        raise ValueError(f"`{model_id_or_path}` is not a local run directory or checkpoint (nothing is fetched from the hub)")
    config_json, stats_json = run_dir / "config.json", run_dir / "dataset_statistics.json"
    assert config_json.exists(), f"Missing `config.json` for `{run_dir = }`"
    assert stats_json.exists(), f"Missing `dataset_statistics.json` for `{run_dir = }`"
    model_cfg = json.loads(config_json.read_text())
    vla_cfg = model_cfg.get("vla", {})
    base = model_cfg.get("model", {})
    vision_id = base.get("vision_backbone_id", "dinosiglip-vit-so-224px")
    llm_id = base.get("llm_backbone_id", "llama2-7b-pure")
    norm_stats = json.loads(stats_json.read_text())
    vision_backbone, _ = get_vision_backbone_and_transform(vision_id, base.get("image_resize_strategy", "resize-naive"))
    llm_backbone, tok = get_llm_backbone_and_tokenizer(llm_id, llm_max_length=base.get("llm_max_length", 2048),
                                                       hf_token=hf_token, inference_mode=not load_for_training, tokenizer=tokenizer)
    action_tokenizer = ActionTokenizer(tok) if tok is not None else None
    vla = OpenVLA(vla_cfg.get("base_vlm", base.get("model_id", "openvla")), vision_backbone, llm_backbone,
                  arch_specifier=base.get("arch_specifier", "no-align+fused-gelu-mlp"), norm_stats=norm_stats,
                  action_tokenizer=action_tokenizer, device=device, dims=dims)
    vla.load_model_state_dicts(torch.load(checkpoint_pt, map_location="cpu", weights_only=True)["model"])
    return vla


def save_run_dir(vla, run_dir: Union[str, Path], step: int = 0, epoch: int = 0, loss: Optional[float] = None) -> Path:
    """Write the native layout `load_vla` reads (what train.py:139-147 + fsdp.py:95-133 leave behind)."""
    d = Path(run_dir)
    (d / "checkpoints").mkdir(parents=True, exist_ok=True)
    cfg = {"model": {"model_id": vla.model_id, "vision_backbone_id": vla.vision_backbone.identifier,
                     "llm_backbone_id": vla.llm_backbone.identifier, "arch_specifier": vla.arch_specifier,
                     "image_resize_strategy": vla.vision_backbone.image_resize_strategy,
                     "llm_max_length": vla.llm_backbone.llm_max_length}, "vla": {"base_vlm": vla.model_id}}
    (d / "config.json").write_text(json.dumps(cfg, indent=2))
    (d / "dataset_statistics.json").write_text(json.dumps(getattr(vla, "norm_stats", {}) or {}, default=_json_default, indent=2))
    tag = "inf" if loss is None else f"{loss:.4f}"
    path = d / "checkpoints" / f"step-{step:06d}-epoch-{epoch:02d}-loss={tag}.pt"
    torch.save({"model": {m: {k: v.cpu() for k, v in sd.items()} for m, sd in vla.model_state_dicts().items()}}, path)
    return path
