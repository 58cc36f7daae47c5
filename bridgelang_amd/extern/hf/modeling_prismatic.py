"""`OpenVLAForActionPrediction` on MI355X: the reference's HF-style model surface over the HIP engine.

Drop-in for prismatic/extern/hf/modeling_prismatic.py: same class names, `forward()` keyword list and output dataclass
(:291-304, :162-173), `predict_action(input_ids, unnorm_key, **kwargs)` (:506-536), `get_action_dim` /
`get_action_stats` / `_check_unnorm_key` (:538-562) and the attributes callers read (`norm_stats`, `bins`,
`bin_centers`, `vocab_size`, `config.image_sizes`, `vision_backbone.featurizer.patch_embed.num_patches` —
finetune.py:219,270). What sits behind it is not timm/transformers/flash-attn but bridgelang_amd.engine (hand-written
gfx950 kernels); there is no eager/PyTorch fallback, so constructing the model without a GPU + built library raises.

The classes are real `transformers.PreTrainedModel` subclasses, so the reference's callers load them the way they load
the reference (`AutoConfig.register` / `AutoModelForVision2Seq.register` + `from_pretrained(local_dir, torch_dtype=
torch.bfloat16, low_cpu_mem_usage=True, trust_remote_code=True)`, finetune.py:151-166, deploy.py:66-75,
run_openvla_demo.py:21-28 — `register_auto_classes()` below; transformers >= 5 renamed the Auto class to
`AutoModelForImageTextToText`, both are served). `from_pretrained` reads the safetensors shards straight into the packed
weight arena on the GPU (no nn.Parameter copies of the 15 GB checkpoint); `.to()` / `.eval()` / `.device` / `.dtype`
behave as on any HF model as long as the target is the GPU the weights already live on and bf16. The model owns no
nn.Parameters (the weights are the arena), so autograd-based wrappers (PEFT, DDP) have nothing to hook: training goes
through bridgelang_amd.training (vla-scripts/finetune.py, vla-scripts/train.py keep the reference scripts' flags).

Differences from the reference, all deliberate:
  * batched `predict_action` is supported (returns [B, 7]); batch 1 returns the reference's 1-D array.
  * batched generation takes RIGHT-padded prompts + attention_mask (the collator's layout); every sequence gets the ids
    it gets alone (the reference asserts batch 1 in its cached branch, :326, :460-463).
  * the greedy loop runs on-device (no per-token host sync); `do_sample=True` is rejected.
  * when 29871 is appended to the prompt the attention mask is extended with it (reference quirk C.1).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch
from transformers import PreTrainedModel

from ...engine import OpenVLAEngine
from ...weights import TowerDims, VLADims, VLAWeights, allocate
from .configuration_prismatic import OpenVLAConfig, PrismaticConfig

IGNORE_INDEX = -100

# timm id → tower geometry (timm 0.9.10 model definitions)
_TIMM_GEOMETRY = {
    "vit_large_patch14_reg4_dinov2.lvd142m": dict(dim=1024, depth=24, heads=16, mlp=4096, n_prefix=5, layerscale=True),
    "vit_so400m_patch14_siglip_224": dict(dim=1152, depth=27, heads=16, mlp=4304, n_prefix=0, layerscale=False),
}


def dims_from_config(config: PrismaticConfig) -> VLADims:
    if not config.use_fused_vision_backbone or len(config.timm_model_ids) != 2:
        raise NotImplementedError("only the fused DINOv2+SigLIP backbone is on the MI355X path")
    a, b = (_TIMM_GEOMETRY[i] for i in config.timm_model_ids)
    tc = config.text_config
    if tc.num_key_value_heads not in (None, tc.num_attention_heads):
        raise NotImplementedError("grouped-query attention is not on the Llama-2 path")
    return VLADims(dino=TowerDims("vision_backbone.featurizer", chan0=0, **a),
                   siglip=TowerDims("vision_backbone.fused_featurizer", chan0=3, **b),
                   llm_dim=tc.hidden_size, llm_layers=tc.num_hidden_layers, llm_heads=tc.num_attention_heads,
                   llm_inter=tc.intermediate_size, vocab=tc.vocab_size, rms_eps=tc.rms_norm_eps,
                   rope_theta=float(getattr(tc, "rope_theta", None) or 10000.0),
                   max_pos=min(tc.max_position_embeddings, config.llm_max_length), name=config.llm_backbone_id)


@dataclass
class PrismaticCausalLMOutputWithPast:
    """Field-for-field the reference's output class (modeling_prismatic.py:162-173)."""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Any] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor]] = None
    projector_features: Optional[torch.Tensor] = None

    def to_tuple(self) -> tuple:
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions,
                                 self.projector_features) if v is not None)


class PrismaticPreTrainedModel(PreTrainedModel):
    """modeling_prismatic.py:176-213: the HF base class of the reference's models (config class, support flags)."""
    config_class = PrismaticConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = False
    _no_split_modules = ["PrismaticProjector"]
    _skip_keys_device_placement = "past_key_values"
    _supports_flash_attn_2 = True      # attention is this package's own kernel: every attn_implementation request is
    _supports_flash_attn = True        # accepted and means the same thing
    _supports_sdpa = True
    _supports_attention_backend = True

    def _init_weights(self, module) -> None:   # weights come from a checkpoint or init_synthetic(); nothing to initialise
        return


class PrismaticForConditionalGeneration(PrismaticPreTrainedModel):
    config_class = PrismaticConfig
    _ENGINE_CACHE = 8          # captured engines kept per model (LRU): each holds KV caches + activations + a HIP graph

    def __init__(self, config: PrismaticConfig, device: Union[str, torch.device, None] = None,
                 dims: Optional[VLADims] = None) -> None:
        super().__init__(config)
        if config.use_fused_vision_backbone is None:
            raise ValueError("Missing config field `use_fused_vision_backbone`")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._device = torch.device(device)
        self.dims = dims if dims is not None else dims_from_config(config)
        self.weights: VLAWeights = allocate(self.dims, self._device)
        self.vocab_size = self.dims.vocab
        self.pad_token_id = config.pad_token_id
        self._engines: "OrderedDict[tuple, OpenVLAEngine]" = OrderedDict()
        self._forward_engines: "OrderedDict[tuple, OpenVLAEngine]" = OrderedDict()
        # attribute path read by finetune.py:270
        self.vision_backbone = SimpleNamespace(
            featurizer=SimpleNamespace(patch_embed=SimpleNamespace(num_patches=self.dims.n_patches)),
            embed_dim=self.dims.vision_dim)
        self.post_init()

    # ---- HF plumbing over the arena ----
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return torch.bfloat16

    def to(self, *args, **kwargs):
        """nn.Module.to for a model whose weights are a packed bf16 arena on one GPU: a no-op for the device / dtype it
        already has (what `from_pretrained(...).to("cuda:0")` asks for), an error otherwise — load with `device=` instead."""
        device, dtype, _, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype is not None and dtype != torch.bfloat16:
            raise NotImplementedError(f"the MI355X path computes in bfloat16; .to({dtype}) is not available")
        if device is not None:
            want = torch.device(device)
            if want.type == "cuda" and want.index is None:
                want = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
            if want != self._device:
                raise NotImplementedError(f"weights live on {self._device}; pass device={want!s} to from_pretrained() / the "
                                          f"constructor instead of moving the arena")
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device) if isinstance(device, int) else (device or "cuda"))

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("this class is the inference surface; training runs through bridgelang_amd.training "
                                      "(vla-scripts/train.py, vla-scripts/finetune.py)")
        return super().train(False)

    # ---- weights ----
    def init_synthetic(self, seed: int = 0, recipe: str = "init") -> "PrismaticForConditionalGeneration":
        self.weights.fill_synthetic(seed, recipe)
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        self.weights.load_state_dict(state_dict, strict=strict)
        return self

    def state_dict(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        """HF-named tensors (unpacked copies), incl. the never-executed ones the reference loads strictly."""
        return self.weights.state_dict()

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config: Optional[PrismaticConfig] = None,
                        device: Union[str, torch.device, None] = None, device_map: Any = None,
                        dims: Optional[VLADims] = None, torch_dtype: Any = None, dtype: Any = None, **kwargs: Any):
        """LOCAL HF export (`config.json` + `*.safetensors` [+ index], convert_openvla_weights_to_hf.py:235-249) →
        model on the GPU. Shards are streamed one at a time straight into the packed arena; nothing is fetched from the
        hub; files are read with safetensors / json only. Accepted and ignored for call-site compatibility:
        `trust_remote_code`, `low_cpu_mem_usage`, `attn_implementation`, `quantization_config=None`, …"""
        from ...models.load import load_hf_directory
        for k in (torch_dtype, dtype):
            if k not in (None, torch.bfloat16, "bfloat16", "auto"):
                raise NotImplementedError(f"only torch_dtype=torch.bfloat16 is available on the MI355X path (got {k})")
        if kwargs.get("quantization_config") is not None:
            raise NotImplementedError("bitsandbytes quantisation is a CUDA path; not available here")
        if device is None and isinstance(device_map, (str, int, torch.device)) and device_map not in ("auto", "balanced"):
            device = device_map
        return load_hf_directory(cls, pretrained_model_name_or_path, config=config, device=device, dims=dims)

    def save_pretrained(self, save_directory, *args, max_shard_size: Union[int, str] = 5 << 30, **kwargs) -> None:
        """`config.json` + sharded safetensors under HF names (reference callers: finetune.py:322-337)."""
        from ...models.load import save_pretrained as _save
        if isinstance(max_shard_size, str):
            num, unit = float(max_shard_size[:-2]), max_shard_size[-2:].upper()
            max_shard_size = int(num * {"KB": 1 << 10, "MB": 1 << 20, "GB": 1 << 30}[unit])
        _save(self, save_directory, max_shard_bytes=int(max_shard_size))

    def _lru(self, cache: "OrderedDict", key: tuple, build):
        """Engines are heavy (KV caches, activations, a captured graph): keep the most recently used few per model and
        drop the rest, so a server seeing many (batch, prompt length) pairs cannot run HBM dry."""
        eng = cache.get(key)
        if eng is None:
            eng = cache[key] = build()
            while len(cache) > self._ENGINE_CACHE:
                cache.popitem(last=False)
        else:
            cache.move_to_end(key)
        return eng

    def engine(self, batch: int, prompt_len: int, n_new: int = 7, padded: bool = False, cached: bool = False) -> OpenVLAEngine:
        """`cached=True`: the engine behind a `forward(..., use_cache=True)` KV-cache handle — keyed apart from the engines
        `predict_action` / `generate` use, so an interleaved action prediction of the same shape does not invalidate the
        caller's cache (each kind has its own LRU slot; a second cached prefill of the same shape still does)."""
        fp8 = bool(getattr(self, "fp8", False))       # `model.fp8 = True`: W8A8 e4m3 Llama prefill projections (extension)
        return self._lru(self._engines, (batch, prompt_len, n_new, fp8, padded, cached),
                         lambda: OpenVLAEngine(self.weights, batch, prompt_len, n_new=n_new, fp8=fp8 and not padded, padded=padded))

    # ---- forward: the reference's three branches (modeling_prismatic.py:322-415) ----
    cache_new_tokens = 7       # tokens a `forward(..., use_cache=True)` KV cache is sized for (prefill token + 6 cached steps)

    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.LongTensor] = None, attention_mask: Optional[torch.Tensor] = None,
                pixel_values: Optional[torch.FloatTensor] = None, labels: Optional[torch.LongTensor] = None,
                inputs_embeds: Optional[torch.FloatTensor] = None, past_key_values: Optional[Any] = None,
                use_cache: Optional[bool] = None, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, output_projector_features: Optional[bool] = None,
                return_dict: Optional[bool] = None) -> Union[Tuple, PrismaticCausalLMOutputWithPast]:
        """The reference's forward (modeling_prismatic.py:291-447), branch for branch:

          * cached generation — `input_ids [B, 1]` + `past_key_values` (:325-341; the batch-1 assert is lifted): one
            decode step on the KV cache the handle names → logits [B, 1, vocab], the same handle one token longer;
          * language-only forward — `pixel_values is None` (:343-359): fp32 logits [B, L, vocab] (+ loss); `inputs_embeds`
            [B, L, D] may replace `input_ids` here (extension; the reference's own `inputs_embeds` plumbing is dead code,
            SURVEY App. C.3);
          * multimodal forward (:362-415): fp32 logits [B, 256+L, vocab] and, with `labels`, the shifted mean
            cross-entropy over the valid tokens (HF CausalLM loss). With `use_cache=True` (and no labels) it is the FIRST
            STEP OF GENERATION: the engine's generation plan runs — bit for bit the launches `generate()` replays — and
            returns the last position's logits [B, 1, vocab] (what GenerationMixin consumes; `logits_to_keep=1` in
            current transformers) plus an opaque `past_key_values` handle bound to that engine's KV caches.

        `output_hidden_states=True` (all-position branches) returns HF's tuple: embeddings, every decoder layer's output,
        the last one after the final norm. `output_attentions` cannot be served: the attention kernels never materialise
        the probability matrix (neither does the reference under flash-attn). No gradients flow through this class."""
        if output_attentions:
            raise NotImplementedError("output_attentions: the fused attention kernels do not materialise attention weights")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("Invalid PrismaticForConditionalGeneration `forward()` call: need input_ids (or inputs_embeds)")
        from ...forward_full import EngineKVCache, forward_all_rows, forward_cached_step, forward_prefill_cached
        hidden = proj = loss = None
        cache: Optional[EngineKVCache] = None
        if input_ids is not None and input_ids.shape[1] == 1:
            assert past_key_values is not None, "You must provide `past_key_values` during cached generation!"
            assert labels is None, "Unexpected key `labels` provided during cached generation!"
            logits = forward_cached_step(self, input_ids, past_key_values)
            cache = past_key_values
        elif pixel_values is None:
            assert past_key_values is None, "Unexpected key `past_key_values` provided during language-only forward!"
            loss, logits, _, hidden = forward_all_rows(self, input_ids, attention_mask, None, labels, inputs_embeds,
                                                      bool(output_hidden_states))
        else:
            if input_ids is None or input_ids.shape[0] != pixel_values.shape[0]:
                raise ValueError("Non-homogenous batch of (text, image) input -- forward() does not support mixed batches!")
            assert past_key_values is None, "Unexpected key `past_key_values` provided during language-only forward!"
            if use_cache and labels is None and not output_hidden_states and not output_projector_features:
                logits, cache = forward_prefill_cached(self, input_ids, attention_mask, pixel_values, self.cache_new_tokens)
            else:
                loss, logits, proj, hidden = forward_all_rows(self, input_ids, attention_mask, pixel_values, labels,
                                                             inputs_embeds, bool(output_hidden_states))
        out = PrismaticCausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache, hidden_states=hidden,
                                              projector_features=proj if output_projector_features else None)
        if return_dict is False:
            return out.to_tuple()
        return out

    def prepare_inputs_for_generation(self, input_ids: Optional[torch.Tensor] = None, past_key_values: Optional[Any] = None,
                                      inputs_embeds: Optional[torch.FloatTensor] = None,
                                      pixel_values: Optional[torch.FloatTensor] = None,
                                      attention_mask: Optional[torch.Tensor] = None, **kwargs: Any) -> Dict[str, Any]:
        """modeling_prismatic.py:450-485 (GenerationMixin hook): with a cache only the newest token is fed; pixel values,
        mask, cache and `use_cache` ride along. The batch-1 restriction of the reference (:460-463) is lifted, and its
        dead `"input_embeds"` key is spelled `inputs_embeds`."""
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs: Dict[str, Any] = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"attention_mask": attention_mask, "pixel_values": None if past_key_values is not None else pixel_values,
                             "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache")})
        return model_inputs

    # nn.Module.__call__ dispatches to forward()

    # ---- greedy generation ----
    @torch.no_grad()
    def generate(self, input_ids: torch.LongTensor, max_new_tokens: int = 7, pixel_values: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, do_sample: bool = False, use_cache: bool = True,
                 **_: Any) -> torch.LongTensor:
        """Greedy decoding, returns [B, L + max_new_tokens] like GenerationMixin (prompt ‖ new tokens). `use_cache` is
        accepted and ignored: the KV cache is always used (use_cache=False in the fork's demo re-runs the vision towers
        7 times, run_openvla_demo.py:43 — same result, 7× the work). A batch whose attention_mask has zeros is taken as
        RIGHT-padded prompts: pads are hidden from attention and every sequence continues from its own last token."""
        if do_sample:
            raise NotImplementedError("only greedy decoding (do_sample=False) is on the HIP path")
        if pixel_values is None:
            raise ValueError("generate() needs pixel_values")
        B, L = input_ids.shape
        ids, pv = input_ids.to(self.device), pixel_values.to(self.device)
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            eng = self.engine(B, L, max_new_tokens, padded=True)
            eng.set_padded_inputs(ids, pv, attention_mask)
            eng.run_eager()
            new = eng.gen_ids.t()
        else:
            new = self.engine(B, L, max_new_tokens).generate(ids, pv)
        return torch.cat([ids, new], dim=1)


class OpenVLAForActionPrediction(PrismaticForConditionalGeneration):
    config_class = OpenVLAConfig

    def __init__(self, config: OpenVLAConfig, device: Union[str, torch.device, None] = None,
                 dims: Optional[VLADims] = None) -> None:
        super().__init__(config, device, dims)
        self.norm_stats = config.norm_stats
        self.bins = np.linspace(-1, 1, config.n_action_bins)
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0
        # de-tokenisation vocabulary: the padded "multiple of 64" rows are not action tokens (reference :503-504)
        self.vocab_size = self.dims.vocab - config.pad_to_multiple_of

    def predict_action(self, input_ids: Optional[torch.LongTensor] = None, unnorm_key: Optional[str] = None,
                       **kwargs: Any) -> np.ndarray:
        """ids → 7 greedy action tokens → bin centres → un-normalised 7-DoF action (reference :506-536)."""
        input_ids = input_ids.to(self.device)
        m = kwargs.get("attention_mask")
        if m is not None and not bool(m.bool().all()):
            # right-padded batch: the empty token goes behind each sequence's last REAL token (one more column)
            m = m.to(self.device).long()
            n = m.sum(dim=1)
            rows = torch.arange(input_ids.shape[0], device=self.device)
            need = input_ids[rows, n - 1] != 29871
            ids = torch.cat((input_ids, torch.full_like(input_ids[:, :1], self.pad_token_id)), dim=1)
            m = torch.cat((m, torch.zeros_like(m[:, :1])), dim=1)
            ids[rows[need], n[need]] = 29871
            m[rows[need], n[need]] = 1
            input_ids, kwargs["attention_mask"] = ids, m
        else:
            input_ids = self.with_empty_token(input_ids)
            if m is not None and m.shape[1] != input_ids.shape[1]:
                m = m.to(self.device)
                kwargs["attention_mask"] = torch.cat((m, torch.ones_like(m[:, :1])), dim=1)
        n = self.get_action_dim(unnorm_key)
        generated = self.generate(input_ids, max_new_tokens=n, **kwargs)
        return self.actions_from_token_ids(generated[:, -n:].cpu().numpy(), unnorm_key)

    def with_empty_token(self, input_ids: torch.LongTensor) -> torch.LongTensor:
        """Append the special empty token 29871 the Llama tokenizer would have put after "Out:" (reference :510-515)."""
        if torch.all(input_ids[:, -1] == 29871):
            return input_ids
        tail = torch.full((input_ids.shape[0], 1), 29871, dtype=torch.long, device=input_ids.device)
        return torch.cat((input_ids, tail), dim=1)

    def actions_from_token_ids(self, token_ids: np.ndarray, unnorm_key: Optional[str] = None) -> np.ndarray:
        """Generated action token ids [B, n] → un-normalised actions (reference :520-536)."""
        discretized = np.clip(self.vocab_size - token_ids - 1, a_min=0, a_max=self.bin_centers.shape[0] - 1)
        normalized = self.bin_centers[discretized]
        stats = self.get_action_stats(unnorm_key)
        mask = stats.get("mask", np.ones_like(stats["q01"], dtype=bool))
        hi, lo = np.array(stats["q99"]), np.array(stats["q01"])
        actions = np.where(mask, 0.5 * (normalized + 1) * (hi - lo) + lo, normalized)
        return actions[0] if actions.shape[0] == 1 else actions

    @staticmethod
    def _check_unnorm_key(norm_stats: Dict[str, Dict[str, Any]], unnorm_key: Optional[str]) -> str:
        if unnorm_key is None:
            assert len(norm_stats) == 1, (
                f"Your model was trained on more than one dataset, please pass a `unnorm_key` from the following "
                f"options to choose the statistics used for un-normalizing actions: {norm_stats.keys()}")
            unnorm_key = next(iter(norm_stats.keys()))
        assert unnorm_key in norm_stats, (
            f"The `unnorm_key` you chose is not in the set of available dataset statistics, please choose from: "
            f"{norm_stats.keys()}")
        return unnorm_key

    def get_action_dim(self, unnorm_key: Optional[str] = None) -> int:
        return len(self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]["q01"])

    def get_action_stats(self, unnorm_key: Optional[str] = None) -> Dict[str, Any]:
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]


def register_auto_classes() -> None:
    """What finetune.py:151-154 / deploy.py:66-69 do before `from_pretrained`: make `model_type == "openvla"` resolve to
    these classes through the HF Auto classes. transformers < 5 has `AutoModelForVision2Seq`; >= 5 renamed it to
    `AutoModelForImageTextToText` — whichever exist are registered. Idempotent."""
    import transformers
    from transformers import AutoConfig
    for model_type, cfg in (("prismatic", PrismaticConfig), ("openvla", OpenVLAConfig)):
        try:
            AutoConfig.register(model_type, cfg)
        except ValueError:
            pass                                     # already registered
    for name in ("AutoModelForVision2Seq", "AutoModelForImageTextToText"):
        auto = getattr(transformers, name, None)
        if auto is None:
            continue
        for cfg, cls in ((PrismaticConfig, PrismaticForConditionalGeneration), (OpenVLAConfig, OpenVLAForActionPrediction)):
            try:
                auto.register(cfg, cls)
            except ValueError:
                pass
