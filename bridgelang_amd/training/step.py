"""One optimisation step of OpenVLA training on one MI355X: forward with saved activations, hand-written backward,
global-norm clip and AdamW — the device side of the reference's `TrainingStrategy.run_vla_training` loop body
(prismatic/training/strategies/base_strategy.py:296-346) with the module freezing of `PrismaticVLM.freeze_backbones`
(prismatic/models/vlms/prismatic.py:129-241) and the optimizer grouping of `FSDPStrategy.run_setup`
(prismatic/training/strategies/fsdp.py:195-236).

MI355X-first choices (DESIGN.md "Training step"):
  * no autograd, no tracing: the step is three static lists of prepared C calls (forward, backward, repack) over
    preallocated buffers, so each list replays under a HIP graph;
  * no activation checkpointing (the reference enables it for 80 GB parts, fsdp.py:176-186): every layer's activations
    stay resident — ≈ 4.2 MB / token at 7B, 20 GB for 16 × 290 tokens out of 288 GB;
  * weights live twice in bf16, fragment-major for the forward GEMM ([N,K] packed) and transposed for the dgrad GEMM
    ([K,N] packed); wgrad is the same NT GEMM kernel over transposed, token-padded activations, accumulating in fp32
    straight into the flat fp32 gradient buffer;
  * fp32 master weights and AdamW moments are flat buffers, so the optimizer is a handful of launches; the clip
    coefficient is consumed on the device (no host sync inside the step).
"""
from __future__ import annotations

import contextlib

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch

from .. import ops, train_ops as T
from ..engine import rope_tables
from ..ops import (EPI_BIAS, EPI_BIAS_GELU_KEEP, EPI_F32, EPI_F32_BF16R, EPI_GELU_BWD, EPI_NONE, EPI_RES, EPI_SWIGLU_BWD,
                   EPI_SWIGLU_KEEP, Op)
from ..weights import PackedGroup, VLAWeights, _block_view, _unpack
from .sharding import ShardComm, ShardLayout, bucket_key, comm_order

IGNORE_INDEX = -100
WGRAD_NT = os.environ.get("BL_WGRAD_NT", "") not in ("", "0")     # A/B aid: weight gradients through transposed copies
WGRAD_FP8_5PASS = os.environ.get("BL_WGRAD_FP8_5PASS", "") not in ("", "0")   # A/B aid: round 3's five passes per e4m3 wgrad operand pair
# SwiGLU / GELU forward ("f") and backward ("b") as GEMM epilogues (BL_EPI_*_KEEP / BL_EPI_*_BWD) instead of separate
# elementwise passes. ON by default since the tile GEMMs' whole-tile epilogue (round 4, gemm_common.h::epilogue_tile): same box,
# alternating runs at 7B, B = 32: separate passes 409.8 ms / step, forward fused 408.7, backward fused 409.0, both 406.6. With the
# per-store epilogue of rounds 1-3 the fused forms LOST 2-4 ms (442.2 vs 444.5-446.2): one workgroup per CU, so the extra
# stores / exp / erf sat exposed in an epilogue that already cost more than the HBM-bound pass they remove.
# BL_TRAIN_FUSED_ACT=0 (or any string without "f" / "b") selects the separate passes. Bit-identical either way (tested).
_FA = os.environ.get("BL_TRAIN_FUSED_ACT", "fb")
UNFUSED_FWD, UNFUSED_BWD = "f" not in _FA, "b" not in _FA

# stage → (vision trainable, projector trainable, llm: "all" | "last" | "none")   — prismatic.py:129-241
STAGES: Dict[str, Tuple[bool, bool, str]] = {
    "align": (False, True, "none"),
    "finetune": (False, True, "all"), "vla-train": (False, True, "all"),
    "full-finetune": (True, True, "all"), "vla-full-train": (True, True, "all"),
    "last-layer-finetune": (False, False, "last"), "vla-last-layer-train": (False, False, "last"),
    "vla-sandwich-train": (True, True, "last"),
    "lora": (False, False, "none"),          # vla-scripts/finetune.py: base frozen, adapters train (training/lora.py)
}


def trainable_names(w: VLAWeights, stage: str) -> List[str]:
    """HF tensor names with requires_grad=True after `freeze_backbones(stage)`; "last" = embed_tokens, the final decoder
    layer and lm_head (llama2.py:101-102; the final norm is not in that tuple and stays frozen)."""
    if stage not in STAGES:
        raise ValueError(f"Stage `{stage}` is not supported")
    vision, proj, llm = STAGES[stage]
    last = f"language_model.model.layers.{w.dims.llm_layers - 1}."
    out = []
    for name in w.placements:
        if name.startswith("vision_backbone."):
            ok = vision
        elif name.startswith("projector."):
            ok = proj
        elif llm == "all":
            ok = True
        elif llm == "last":
            ok = name.startswith(last) or name in ("language_model.model.embed_tokens.weight", "language_model.lm_head.weight")
        else:
            ok = False
        if ok:
            out.append(name)
    return out


def pl_is_plain(w: VLAWeights, name: str) -> bool:
    return w.placements[name].group is None


# buckets whose PARAMETERS are sharded under fsdp-full-shard: every FSDP unit of the reference — decoder layers, ViT blocks
# (+ patch embeddings), projector, and the root's token embeddings / lm_head (prismatic.py:285-306, dinosiglip_vit.py:136-140,
# base_llm.py:182-188, fsdp.py:160-168). Not sharded: the two buckets of small plain tensors.
def param_sharded_key(key: str) -> bool:
    return key.startswith(("llm.layer", "vision.", "projector", "llm.lm_head", "llm.embed"))


def pool_of(key: str) -> str:
    """Gather-slot pool of a parameter-sharded bucket: uniform decoder layers / the vision units / the three big singletons."""
    return "layers" if key.startswith("llm.layer") else "vision" if key.startswith("vision.") else "head"


def no_decay(name: str, shape: Tuple[int, ...]) -> bool:
    """fsdp.py:208: `param.ndim <= 1 or name.endswith(".bias")` → weight_decay 0."""
    return len(shape) <= 1 or name.endswith(".bias")


@dataclass
class Unit:
    """One optimizer unit: a packed GEMM weight group (logical [n, k] matrix) or a plain tensor."""
    key: str
    offset: int                 # into the flat (bucket-padded) parameter space
    numel: int
    group: Optional[PackedGroup]
    dst: Optional[torch.Tensor]        # plain: contiguous bf16 view of the live parameter
    decay: bool
    bucket: int
    names: Tuple[str, ...] = ()


class ParamStore:
    """Trainable tensors as one flat parameter space (sharding.ShardLayout): full fp32 gradients and a full bf16 staging
    copy on every rank, fp32 masters and AdamW moments for this rank's 1/world slice of every bucket only."""

    def __init__(self, w: VLAWeights, stage: str, world: int = 1, rank: int = 0,
                 extra: Optional[List[Tuple[str, torch.Tensor, bool, str]]] = None, only: Optional[Sequence[str]] = None,
                 shard_params: bool = False, defer_grads: bool = False):
        """`extra`: additional trainable bf16 tensors outside the model's own table — (name, flat live tensor, decayed,
        bucket key), e.g. LoRA adapters. `only`: restrict the stage's trainable set to these HF names (whole fused groups;
        diagnostics / tests: optimizer state for a handful of tensors instead of the model). `shard_params`: the
        decoder-layer buckets keep NO full bf16 copy — each rank holds its 1/world slice (`own`), the unit is gathered
        around its use (FSDP FULL_SHARD, fsdp.py:84-87; TrainStep(shard_params=True))."""
        self.w, self.stage = w, stage
        specs = w._specs()
        names = set(trainable_names(w, stage))
        if only is not None:
            keep = set(only)
            for g in w.groups:
                if keep & set(g.members):
                    keep |= set(g.members)
            names &= keep
        self.names = names
        self.units: List[Unit] = []
        self.by_name: Dict[str, Unit] = {}
        self.layout = lay = ShardLayout(world, rank)
        cur = None
        for gi, g in enumerate(w.groups):                       # GEMM weight groups (all decayed: ndim >= 2)
            tr = [n in names for n in g.members]
            if not any(tr):
                continue
            assert all(tr), f"group {g.members} is only partly trainable"
            key = bucket_key(g.members[0])
            if key != cur:
                lay.begin(key, True)
                cur = key
            u = Unit(f"group{gi}", 0, g.n * g.k, g, None, True, len(lay.buckets) - 1, tuple(g.members))
            u.offset = lay.add(u.numel, len(self.units))
            self.units.append(u)
            for n in g.members:
                self.by_name[n] = u
        EMBED = "language_model.model.embed_tokens.weight"
        # (bucket key, decayed, names): the token embeddings are half of the reference's root FSDP unit (fsdp.py:160-168) and get
        # a bucket of their own right behind the GEMM-weight buckets (parameter-sharded with them); the other plain tensors
        # (norm scales, biases, LayerScale, position / class tokens: 0.1 % of the model) share one bucket per weight-decay class
        plain_buckets = [("llm.embed", True, [EMBED])] if EMBED in names and pl_is_plain(w, EMBED) else []
        for decay in (True, False):
            plain_buckets.append(("plain.decay" if decay else "plain.nodecay", decay,
                                  [n for n, pl in w.placements.items() if pl.group is None and n in names and n != EMBED
                                   and (not no_decay(n, specs[n].shape)) == decay]))
        for bkey, decay, members in plain_buckets:
            todo = [(n, w.placements[n]) for n in members]
            if not todo:
                continue
            lay.begin(bkey, decay)
            for name, pl in todo:
                assert pl.ld == pl.cols or pl.rows == 1
                dst = pl.dst.view(-1)[pl.offset:pl.offset + pl.rows * pl.cols]
                u = Unit(name, 0, dst.numel(), None, dst, decay, len(lay.buckets) - 1, (name,))
                u.offset = lay.add(u.numel, len(self.units))
                self.units.append(u)
                self.by_name[name] = u
        cur = None
        for name, dst, decay, bkey in (extra or []):
            if bkey != cur:
                lay.begin(bkey, decay)
                cur = bkey
            assert dst.is_contiguous() and dst.dtype == torch.bfloat16 and lay.buckets[-1].decay == decay
            u = Unit(name, 0, dst.numel(), None, dst.view(-1), decay, len(lay.buckets) - 1, (name,))
            u.offset = lay.add(u.numel, len(self.units))
            self.units.append(u)
            self.by_name[name] = u
        lay.close()
        self.total = lay.total
        self.n_params = sum(u.numel for u in self.units)
        dev = w.embed.device
        # Parameter-sharded buckets (FSDP FULL_SHARD units = the decoder layers) occupy a contiguous range [lo, hi) of the
        # flat space that is CUT OUT of the addressing of the replicated buffers — the bf16 staging copy of the updated
        # parameters and the flat fp32 gradient: for those buckets a rank keeps only its 1/world slice of the parameters
        # (`own`, bf16) and of the reduced gradient (`gshard`, fp32); the full gradient of ONE layer exists transiently in
        # one of two slots (`gslots`) between that layer's weight-gradient GEMMs and its reduce-scatter, as FSDP frees a
        # unit's full gradient after reduce-scatter (fsdp.py:160-168).
        self.shard_params = shard_params
        sh = [b for b in lay.buckets if shard_params and param_sharded_key(b.key)]
        self.sharded_keys = {b.key for b in sh}
        self._cut = (sh[0].offset, sh[-1].offset + sh[-1].numel) if sh else (lay.total, lay.total)
        assert sum(b.numel for b in sh) == self._cut[1] - self._cut[0], "parameter-sharded buckets must be contiguous"
        self._n_rep = n_rep = lay.total - (self._cut[1] - self._cut[0])
        self.stage_bf16 = torch.zeros(max(n_rep, 8), dtype=torch.bfloat16, device=dev)
        self.own_off, n_own = {}, 0
        # gradient slots: per pool two transient fp32 buffers sized for the pool's largest bucket; a bucket uses slot
        # (its index in the pool's forward order) % 2 — for the decoder layers that is layer % 2
        self.grad_slot_of: Dict[str, Tuple[str, int]] = {}
        self._slot_numel: Dict[str, int] = {}
        seen: Dict[str, int] = {}
        for b in sh:
            self.own_off[b.key] = n_own
            n_own += lay.shard_numel(b)
            pool = pool_of(b.key)
            idx = int(b.key[len("llm.layer"):]) if pool == "layers" else seen.get(pool, 0)
            seen[pool] = seen.get(pool, 0) + 1
            self.grad_slot_of[b.key] = (pool, idx % 2)
            self._slot_numel[pool] = max(self._slot_numel.get(pool, 0), b.numel)
        self._n_own = n_own
        self.own = torch.zeros(max(n_own, 8), dtype=torch.bfloat16, device=dev) if sh else None
        self.grad = self.gshard = None
        self.gslots: Dict[str, List[torch.Tensor]] = {}
        self.use_grad_slots = True          # TrainStep clears it when no collective runs (world 1): the "slice" is then the whole
                                            # bucket and the weight-gradient GEMMs write the persistent buffer directly
        if not defer_grads:                 # TrainStep defers: it first gives the model's own decoder-layer allocation back
            self.alloc_grads()
        f = lambda: torch.zeros(max(lay.local_total, 8), dtype=torch.float32, device=dev)
        self.master, self.m, self.v = f(), f(), f()
        for bi, b in enumerate(lay.buckets):                    # masters ← this rank's slice of the live bf16 weights
            full = torch.zeros(b.numel, dtype=torch.float32, device=dev)
            for ui in b.members:
                u = self.units[ui]
                sl = full[u.offset - b.offset:u.offset - b.offset + u.numel]
                if u.group is not None:
                    sl.view(u.group.n, u.group.k).copy_(_unpack(u.group.packed))
                else:
                    sl.copy_(u.dst)
            lo, hi = lay.shard_range(b)
            lo_l = lay.local_offset(b)
            self.master[lo_l:lo_l + hi - lo].copy_(full[lo - b.offset:hi - b.offset])
            if b.key in self.sharded_keys:
                self.own_slice(b).copy_(full[lo - b.offset:hi - b.offset])
            del full
        self.blocks_per_bucket = 2048
        self.partial = torch.zeros(self.blocks_per_bucket * max(len(lay.buckets), 1), dtype=torch.float32, device=dev)
        self.norm_coef = torch.zeros(2, dtype=torch.float32, device=dev)     # [total norm, clip coefficient]
        self.step_count = 0

    def alloc_grads(self) -> None:
        """The fp32 gradient buffers: flat for the replicated buckets; this rank's slices + two transient layer slots for
        the parameter-sharded ones."""
        if self.grad is not None:
            return
        dev = self.w.embed.device
        self.grad = torch.zeros(max(self._n_rep, 8), dtype=torch.float32, device=dev)
        if self.sharded_keys:
            self.gshard = torch.zeros(max(self._n_own, 8), dtype=torch.float32, device=dev)
            if self.use_grad_slots:
                self.gslots = {pool: [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(2)]
                               for pool, n in self._slot_numel.items()}

    # ---- views ----
    def _replicated(self, buf: torch.Tensor, offset: int, numel: int) -> torch.Tensor:
        """View of flat range [offset, offset + numel) in a buffer that skips the parameter-sharded range."""
        lo, hi = self._cut
        assert offset + numel <= lo or offset >= hi, "parameter-sharded buckets have no replicated copy"
        o = offset if offset < lo else offset - (hi - lo)
        return buf[o:o + numel]

    def stage_view(self, offset: int, numel: int) -> torch.Tensor:
        """bf16 staging view of flat range [offset, offset + numel) (never inside the parameter-sharded range)."""
        return self._replicated(self.stage_bf16, offset, numel)

    def grad_range(self, offset: int, numel: int) -> torch.Tensor:
        """fp32 gradient of flat range [offset, offset + numel) of the replicated buckets."""
        return self._replicated(self.grad, offset, numel)

    def own_slice(self, b) -> torch.Tensor:
        """This rank's bf16 slice of a parameter-sharded bucket."""
        o = self.own_off[b.key]
        return self.own[o:o + self.layout.shard_numel(b)]

    def grad_slot(self, b) -> torch.Tensor:
        """The transient full-size fp32 gradient of parameter-sharded bucket b (decoder layer l uses slot l % 2 of the
        layer pool; the vision / head units alternate inside their pools)."""
        if not self.use_grad_slots:          # one rank, no collective: the rank's slice of the reduced gradient is the bucket itself
            assert self.layout.world == 1
            return self.reduced_grad(b)
        pool, parity = self.grad_slot_of[b.key]
        return self.gslots[pool][parity][:b.numel]

    def reduced_grad(self, b) -> torch.Tensor:
        """This rank's slice of bucket b's gradient once reduced: what the norm and AdamW read."""
        if b.key in self.sharded_keys:
            o = self.own_off[b.key]
            return self.gshard[o:o + self.layout.shard_numel(b)]
        lo, hi = self.layout.shard_range(b)
        return self.grad_range(lo, hi - lo)

    def _unit_grad(self, u: Unit) -> torch.Tensor:
        b = self.layout.buckets[u.bucket]
        if b.key in self.sharded_keys:
            return self.grad_slot(b)[u.offset - b.offset:u.offset - b.offset + u.numel]
        return self.grad_range(u.offset, u.numel)

    def grad_view(self, name_or_unit) -> torch.Tensor:
        """fp32 gradient of a unit: [n, k] for a group, flat for a plain tensor (for a parameter-sharded decoder layer:
        the transient slot the layer's weight-gradient GEMMs write — valid until the slot's next layer overwrites it)."""
        u = name_or_unit if isinstance(name_or_unit, Unit) else self.by_name[name_or_unit]
        sl = self._unit_grad(u)
        return sl.view(u.group.n, u.group.k) if u.group is not None else sl

    def _named(self, flat: Optional[torch.Tensor], name: str, unit_slice: Optional[torch.Tensor] = None) -> torch.Tensor:
        u = self.by_name[name]
        sl = unit_slice if unit_slice is not None else flat[u.offset:u.offset + u.numel]
        if name not in self.w.placements:                       # extra unit: flat
            return sl
        pl = self.w.placements[name]
        shape = self.w._specs()[name].shape
        if u.group is None:
            return sl.view(shape)
        return _block_view(sl, pl).reshape(shape)

    def named_grad(self, name: str) -> torch.Tensor:
        """Gradient under its HF name / shape (a copy for grouped tensors): the local, un-reduced gradient of a replicated
        bucket. Parameter-sharded units have no such thing once collectives run — their full gradient lives in a transient
        slot that the flush reduce-scatters in place and later units overwrite — so this raises for them; without
        collectives (one rank) the rank's "slice" is the whole bucket and the call is valid."""
        u = self.by_name[name]
        if self.layout.buckets[u.bucket].key in self.sharded_keys and self.use_grad_slots:
            raise RuntimeError(f"{name}: the full gradient of a parameter-sharded unit is transient (reduce-scattered in its slot, "
                               "overwritten two units later); read store.reduced_grad(bucket) — this rank's reduced slice — instead")
        return self._named(None, name, self._unit_grad(u))

    def full_master(self, comm: Optional[ShardComm] = None) -> torch.Tensor:
        """All fp32 masters in the flat layout (gathered over ranks when sharded) — checkpoints and tests."""
        return (comm or ShardComm(self.layout)).gather_full(self.master)

    def named_master(self, name: str, full: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._named(self.full_master() if full is None else full, name)

    def master_state_dict(self, comm: Optional[ShardComm] = None) -> Dict[str, torch.Tensor]:
        full = self.full_master(comm)
        return {n: self._named(full, n).clone() for n in self.by_name}

    def trainable(self, name: str) -> bool:
        return name in self.by_name

    def unit_of_packed(self, packed: torch.Tensor) -> Optional[Unit]:
        for u in self.units:
            if u.group is not None and u.group.packed.data_ptr() == packed.data_ptr():
                return u
        return None


def transposed_pack(packed: torch.Tensor, n: int, k: int, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Packed [n, k] weight → packed [k, n] (the dgrad operand)."""
    rm = _unpack(packed).contiguous()
    t = torch.empty(k, n, dtype=torch.bfloat16, device=packed.device)
    T.transpose_pad(rm, t, n)
    return ops.pack_weight(t)


class TrainStep:
    """Static-shape training step for batches of B samples with L (padded) text tokens."""

    def __init__(self, weights: VLAWeights, stage: str, batch: int, prompt_len: int, *, max_grad_norm: float = 1.0,
                 weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8, store: Optional[ParamStore] = None,
                 world: int = 1, rank: int = 0, group=None, reduce_dtype: torch.dtype = torch.float32, lora=None,
                 force_comm: bool = False, recompute: bool = False, shard_params: bool = False, fp8: bool = False,
                 fp8_wgrad: bool = False):
        """`lora`: a training.lora.LoraAdapters → stage "lora": the base model is frozen and only the adapters train.
        `recompute`: keep only each decoder layer's input and replay its forward inside the backward pass.
        `shard_params`: FSDP FULL_SHARD for the decoder layers (fsdp.py:84-87) — every rank keeps 1/world of each layer's
        bf16 weights; a layer is all-gathered into one of two slots (and packed into the forward / dgrad layouts there) one
        layer ahead of its forward and again ahead of its backward, on the communication stream; AdamW updates the
        rank's slice in place. The model's own decoder-layer allocation is freed (`materialize_params()` brings it back).
        `fp8`: the decoder layers' forward and input-gradient GEMMs (q/k/v, o, gate/up, down: 2/3 of the step's GEMM work)
        run W8A8 on the e4m3 MFMA path (BASELINE configs[4]) — activations / output gradients quantised per token row,
        weights per output channel (forward) and per input channel (the transposed dgrad copy), fp32 accumulation, bf16
        results; weight gradients stay bf16 x bf16 (TN GEMM), masters and AdamW fp32. No reference counterpart: agreement
        with the bf16 step is to quantisation noise (tests/test_train_step_gpu.py states the bound).
        `fp8_wgrad` (with `fp8`): the decoder layers' WEIGHT-gradient GEMMs run on the e4m3 MFMA path too, so all three GEMM
        families of those layers are fp8: dW[n, k] = Σ_t dy[t, n]·x[t, k] contracts over tokens, so dy and x are transposed
        to token-contiguous rows and quantised per ROW of the transposed matrix — one scale per output channel n of dy and
        per input channel k of x, i.e. along the non-contracted dimensions, where the scales factor out of the sum exactly —
        and dW = s_dy[n]·s_x[k]·(dyT8 · xT8ᵀ) accumulates in fp32 straight into the flat gradient buffer."""
        if (lora is not None) != (stage == "lora"):
            raise ValueError("stage 'lora' and the `lora` adapters go together")
        if (recompute or shard_params or fp8) and lora is not None:
            raise ValueError("activation recomputation / parameter sharding / fp8 GEMMs are planned for the full-parameter stages only")
        if fp8 and (weights.dims.llm_dim % 128 or weights.dims.llm_inter % 128):
            raise ValueError("fp8 GEMMs need llm_dim and llm_inter to be multiples of 128")
        if fp8_wgrad and not fp8:
            raise ValueError("fp8_wgrad extends fp8=True (e4m3 forward / dgrad) to the weight-gradient GEMMs")
        self.fp8, self.fp8_wgrad = fp8, fp8_wgrad
        self.fp8_wgrad_gemms = [0, 0]          # planned weight-gradient GEMMs [on the e4m3 path, fallen back to bf16]
        self.recompute, self.shard_params = recompute, shard_params
        self.lora = lora
        self.train_vision = STAGES[stage][0] or lora is not None      # towers need their training-form forward
        self.w, self.dims, self.stage = weights, weights.dims, stage
        d = self.dims
        self.B, self.L, self.S = batch, prompt_len, prompt_len + d.n_patches
        if self.S > d.max_pos:
            raise ValueError(f"sequence of {self.S} positions exceeds the model's {d.max_pos} (model_max_length)")
        self.max_grad_norm, self.weight_decay, self.betas, self.eps = max_grad_norm, weight_decay, betas, eps
        if store is None:
            store = ParamStore(weights, stage, world, rank, extra=lora.plain_units() if lora is not None else None,
                               shard_params=shard_params, defer_grads=True)
        self.store = store
        assert self.store.stage == stage and self.store.shard_params == shard_params
        shard_params = self.shard_params = shard_params and bool(self.store.sharded_keys)      # frozen LLM: nothing to shard
        st = self.store
        self.comm = ShardComm(st.layout, group, reduce_dtype, force=force_comm)
        self.world = st.layout.world
        dev = weights.embed.device
        self.device = dev
        self._wT: Dict[int, torch.Tensor] = {}       # transposed packed weights for dgrad
        self._w8: Dict[int, dict] = {}              # packed bf16 weight (data_ptr) → its e4m3 forward / dgrad copies + scales
        self._cur_layer = -1                   # decoder layer whose ops are being planned (slot tensors are shared by layers)
        self._cur_unit: Optional[str] = None   # vision / head unit (bucket key) whose ops are being planned
        self._units: Dict[str, dict] = {}      # parameter-sharded vision / head units (see _setup_unit_pools)
        self._materialized = False
        if shard_params:                       # first: the model's layer allocation is given back before anything else is reserved
            self._setup_param_shards()
            self._setup_unit_pools()
        if st.grad is None:
            st.use_grad_slots = self.comm.active
        st.alloc_grads()
        B, S, D, I, V, NL = batch, self.S, d.llm_dim, d.llm_inter, d.vocab, d.llm_layers
        Tn = B * S
        self.T, self.Tp = Tn, (Tn + 63) // 64 * 64
        z = lambda *shape, dtype=torch.bfloat16: torch.zeros(*shape, dtype=dtype, device=dev)
        # LoRA K-concatenation: the input x of an adapted linear and the gradient dy of its output live in buffers that
        # are R columns wider than the tensor; t = x·Aᵀ / dt = dy·(sB) land in those columns, so y = [x | t]·[W | sB]ᵀ
        # and dx = [dy | dt]·[Wᵀ | Aᵀ]ᵀ are single GEMMs over K + R (no separate rank-R update pass over y / dx).
        self._wide: Dict[int, torch.Tensor] = {}
        R_ = lambda packed: (lora.get(packed).R if (lora is not None and lora.get(packed) is not None) else 0)

        def za(rows: int, cols: int, extra: int) -> torch.Tensor:
            if not extra:
                return z(rows, cols)
            wide = z(rows, cols + extra)
            view = wide[:, :cols]
            self._wide[view.data_ptr()] = wide
            return view
        self._za, self._R = za, R_
        L0 = weights.layers[0]
        # ---- inputs ----
        self.input_ids = z(B, prompt_len, dtype=torch.int64)
        self.key_mask = torch.ones(B, S, dtype=torch.uint8, device=dev)
        self.targets = torch.full((Tn,), IGNORE_INDEX, dtype=torch.int64, device=dev)
        # ---- frozen vision front end: reuse the inference engine's tower plans (its LLM buffers are not allocated twice:
        #      prompt_len 1, no decode) ----
        from ..engine import OpenVLAEngine
        self._vis = OpenVLAEngine(weights, batch, 1, n_new=1, vision_only=True)
        if any(info["pool"] == "vision" for info in self._units.values()):
            self._vis.dino_ops = self._vis.siglip_ops = self._vis.vision_ops = []      # the towers' weights live in gather slots
        self.pixel_values = self._vis.pixel_values
        self.feats = self._vis.feats if lora is None else za(B * 256, d.vision_dim, R_(weights.fc1_w))
        # ---- saved activations ----
        Pv = 4 * d.vision_dim
        self.z1, self.z2, self.p3 = z(B * 256, Pv), z(B * 256, D), z(B * 256, D)
        self.p1, self.p2 = za(B * 256, Pv, R_(weights.fc2_w)), za(B * 256, D, R_(weights.fc3_w))
        self.x = [z(Tn, D) for _ in range(NL + 1)]            # residual stream entering layer l (x[NL] = final)
        # per-layer saved activations; with `recompute` (the reference's activation checkpointing of the decoder layers,
        # fsdp.py:171-183) every layer shares ONE set and only the layer inputs x[l] are kept
        def per_layer(make):
            if not recompute:
                return [make() for _ in range(NL)]
            one = make()
            return [one] * NL
        self.xm = per_layer(lambda: z(Tn, D))
        self.h1 = per_layer(lambda: za(Tn, D, R_(L0.qkv_w)))
        self.h2 = per_layer(lambda: za(Tn, D, R_(L0.gu_w)))
        self.qkv = per_layer(lambda: za(Tn, 3 * D, R_(L0.qkv_w)))     # same leading dimension as dqkv (shared strides)
        self.ao = per_layer(lambda: za(Tn, D, R_(L0.o_w)))
        self.gu = per_layer(lambda: z(Tn, 2 * I))
        self.act = per_layer(lambda: za(Tn, I, R_(L0.down_w)))
        pad = (S + 31) // 32 * 32
        self.lse = per_layer(lambda: z(B * d.llm_heads * pad, dtype=torch.float32))
        self.delta = z(B * d.llm_heads * pad, dtype=torch.float32)
        self.hn = z(Tn, D)
        self.logits = z(Tn, V, dtype=torch.float32)
        self.row_loss, self.mean_cnt = z(Tn, dtype=torch.float32), z(2, dtype=torch.float32)
        self.cos, self.sin = rope_tables(d.head_dim, d.max_pos, d.rope_theta, dev)
        # ---- backward scratch ----
        self.dlogits = z(Tn, V)
        rres = max(R_(L0.down_w), R_(L0.o_w))                  # dx / dx2 are the dy of down_proj and o_proj
        self.dxa, self.dxb, self.dh, self.dao = za(Tn, D, rres), za(Tn, D, rres), z(Tn, D), za(Tn, D, R_(L0.o_w))
        self.dqkv, self.dgu, self.dact = za(Tn, 3 * D, R_(L0.qkv_w)), za(Tn, 2 * I, R_(L0.gu_w)), z(Tn, I)
        self.dp3, self.dz2, self.dz1 = za(B * 256, D, R_(weights.fc3_w)), za(B * 256, D, R_(weights.fc2_w)), za(B * 256, Pv, R_(weights.fc1_w))
        self.dp2, self.dp1 = z(B * 256, D), z(B * 256, Pv)
        nmax = max(V, 3 * D, 2 * I, Pv)
        self.tA, self.tB, self.tBp = z(nmax * self.Tp), z(nmax * self.Tp), z(nmax * self.Tp)
        towers = (weights.dino, weights.siglip) if self.train_vision else ()
        mv = max((B * tw.dims.tokens for tw in towers), default=0)
        dv_ = max((tw.dims.dim for tw in towers), default=0)
        self.norm_ws = z(max(((Tn + 15) // 16) * D, 2 * ((mv + 15) // 16) * dv_), dtype=torch.float32)
        self.col_ws = z(max(((Tn + 31) // 32) * max(Pv, D), 256 * dv_, ((mv + 15) // 16) * dv_ * 4), dtype=torch.float32)
        self._frozen_dw = z(2 * max(D, Pv, dv_), dtype=torch.float32)         # sink for vector grads of frozen tensors
        self._lora_t: Dict[int, torch.Tensor] = {}                            # saved t = x·Aᵀ per adapted linear
        self._Wext: Dict[int, torch.Tensor] = {}                              # packed [W | s·B]   per adapted linear
        self._WText: Dict[int, torch.Tensor] = {}                             # packed [Wᵀ | Aᵀ]
        self._BsT: Dict[int, torch.Tensor] = {}                               # packed (s·B)ᵀ: the dt = dy·(sB) operand
        self._adapter_ops: List[Op] = []                                      # refresh of the adapter parts (repack plan)
        self.dfeats = z(B * 256, d.vision_dim) if self.train_vision else None
        self.vis = [self._alloc_tower(tw) for tw in towers]
        # ---- transposed weights for dgrad ----
        self.ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        if fp8:
            self._setup_fp8()
        if lora is not None:
            self._build_extended_weights()
        # execution order of the vision units (forward; the backward visits each tower's blocks top down, then its stem)
        vkey = lambda tw, i: f"vision.{tw.dims.prefix.split('.')[1]}." + ("stem" if i < 0 else f"block{i:02d}")
        self._vkey = vkey
        self._vseq_fwd = [vkey(tw, i) for tw in towers for i in range(-1, tw.dims.n_run)]
        self._vseq_bwd = [vkey(tw, i) for tw in towers for i in list(range(tw.dims.n_run - 1, -1, -1)) + [-1]]
        # The two towers are independent between the pixels and the projector (forward) and between the projector's input
        # gradient and their own weight gradients (backward): with un-sharded parameters the second tower's plans run on a
        # side stream inside the same graph, as in the inference engine — their K ≈ 1 K GEMMs pay ≈ 14 µs of launch + ramp +
        # epilogue per 25 µs main loop and fill 130–270 of 256 CUs. (The backward only on one GPU: with ranks to reduce over
        # it is issued bucket by bucket on one stream, see backward().) The side tower gets its own scratch (split-K
        # workspace, column-sum / norm / adapter-gradient partials); activations and gradients are per tower anyway. With
        # sharded parameters, fp8 or LoRA dropout the plans stay one list. BL_TRAIN_VISION_STREAMS=0: single stream (A/B).
        self._vis2 = (self.train_vision and len(towers) == 2 and not shard_params and not fp8
                      and (lora is None or lora.dropout == 0.0)      # the dropout buffers serve one adapted linear at a time
                      and os.environ.get("BL_TRAIN_VISION_STREAMS", "1") != "0")
        self._vis_stream = torch.cuda.Stream(device=dev) if self._vis2 else None
        self._scratch2 = (dict(ws=torch.empty_like(self.ws), col_ws=torch.empty_like(self.col_ws), norm_ws=torch.empty_like(self.norm_ws))
                          if self._vis2 else None)
        self.vision_forward_ops: List[Op] = []
        self._vfwd_tower: List[List[Op]] = []
        for ti, (tw, sv, col) in enumerate(zip(towers, self.vis, (0, d.dino.dim))):
            with self._side_scratch(ti == 1):
                self._vfwd_tower.append(self._plan_tower_forward(tw, col, sv))
            self.vision_forward_ops += self._vfwd_tower[-1]
        self.forward_ops = self._plan_forward()
        self.tn_ws = z(8 << 20, dtype=torch.float32) if lora is not None else None
        if self._vis2 and self.tn_ws is not None:
            self._scratch2["tn_ws"] = torch.empty_like(self.tn_ws)
        self._ready: List[Tuple[int, str]] = []      # (number of backward ops enqueued, bucket key complete at that point)
        self.backward_ops = self._plan_backward()
        self.repack_ops = self._plan_repack()
        self._graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self._comm_stream = torch.cuda.Stream(device=dev) if (self.comm.active or shard_params) else None
        self._rs_scratch = (z(max(st._slot_numel.values())) if shard_params and self.comm.active
                            and reduce_dtype != torch.float32 else None)        # bf16 wire copy of one unit's gradients

    # ---- helpers ------------------------------------------------------------------------------------------------
    # ---- fp8 (e4m3) forward / dgrad GEMMs of the decoder layers ------------------------------------------------------
    def _setup_fp8(self) -> None:
        d, dev, Tn = self.dims, self.device, self.T
        D, I = d.llm_dim, d.llm_inter
        u8 = lambda *shape: torch.zeros(*shape, dtype=torch.uint8, device=dev)
        self._q8 = {D: u8(Tn, D), I: u8(Tn, I)}                          # quantised GEMM inputs (one linear at a time)
        self._q8_dy = u8(Tn, max(3 * D, 2 * I))                          # quantised output gradients
        self._sx, self._sdy = (torch.zeros(Tn, dtype=torch.float32, device=dev) for _ in range(2))
        if self.fp8_wgrad:                     # token-contiguous (transposed) operands of the e4m3 weight-gradient GEMM
            Tq = (Tn + 127) // 128 * 128       # its contraction length: tokens, padded with zero columns to the MFMA's K = 128
            nmx = max(3 * D, 2 * I)
            self._Tq = Tq
            self._wg_qA = u8(nmx * Tq)                                                 # dyᵀ codes [N, Tq] row-major
            self._wg_pB = u8(I * Tq)                                                   # xᵀ codes [K, Tq] in the fragment-major packing
            self._wg_sA = torch.zeros(nmx, dtype=torch.float32, device=dev)
            self._wg_sB = torch.zeros(I, dtype=torch.float32, device=dev)
            self._wg_amax = torch.zeros(nmx + I, dtype=torch.float32, device=dev)      # column |max| of dy (first N) and x (next K)
            if WGRAD_FP8_5PASS:                # A/B aid: the round-3 form through separate transpose / quantise / pack passes
                self._wg_tA = torch.zeros(nmx * Tq, dtype=torch.bfloat16, device=dev)
                self._wg_tB = torch.zeros(I * Tq, dtype=torch.bfloat16, device=dev)
                self._wg_qB = u8(I * Tq)
        self._fp8_scratch()
        if self.shard_params:                  # sharded layers are quantised in their gather (slots carry the e4m3 copies)
            return
        for lw in self.w.layers:
            for key in self._LAYER_KEYS:
                p = getattr(lw, key)
                n, k = p.shape[0] * 16, p.shape[1] * 32
                self._w8[p.data_ptr()] = e = self._fp8_entry(n, k, dev)
                rm = ops.unpack_weight(p).contiguous()                   # start-up only; later the optimizer's bf16 copy
                ops.run_all(self._fp8_weight_ops(rm, e))
                del rm

    def _fp8_scratch(self) -> None:
        if not hasattr(self, "_q_tmp"):
            D, I = self.dims.llm_dim, self.dims.llm_inter
            nmax = max(3 * D * D, 2 * I * D)
            self._q_tmp = torch.zeros(nmax, dtype=torch.uint8, device=self.device)
            self._t_tmp = torch.zeros(nmax, dtype=torch.bfloat16, device=self.device)

    @staticmethod
    def _fp8_entry(n: int, k: int, dev) -> dict:
        return dict(w8=torch.zeros(n // 16, k // 64, 64, 8, dtype=torch.bfloat16, device=dev),
                    sw=torch.zeros(n, dtype=torch.float32, device=dev),
                    wT8=torch.zeros(k // 16, n // 64, 64, 8, dtype=torch.bfloat16, device=dev),
                    swT=torch.zeros(k, dtype=torch.float32, device=dev))

    def _fp8_weight_ops(self, rm: torch.Tensor, e: dict) -> List[Op]:
        """Logical bf16 weight [n, k] → e4m3 forward copy (scale per output channel n) and e4m3 transposed copy for dgrad
        (scale per input channel k), both in the fragment-major packing bl_gemm_fp8 reads."""
        n, k = rm.shape
        q = self._q_tmp[:n * k].view(n, k)
        qT = self._q_tmp[:n * k].view(k, n)
        t = self._t_tmp[:n * k].view(k, n)
        as_pairs = lambda c: c.view(torch.bfloat16)                    # e4m3 codes as 2-byte units: [r, c] → [r, c / 2]
        return [ops.quantize_rows_fp8(rm, q, e["sw"], run=False)[2], T.pack(as_pairs(q), e["w8"], run=False),
                T.transpose_pad(rm, t, n, run=False), ops.quantize_rows_fp8(t, qT, e["swT"], run=False)[2],
                T.pack(as_pairs(qT), e["wT8"], run=False)]

    # ---- parameter sharding (FSDP FULL_SHARD for the decoder layers) ---------------------------------------------------
    _LAYER_KEYS = ("qkv_w", "o_w", "gu_w", "down_w")

    def _setup_param_shards(self) -> None:
        """Two gather slots (logical bf16 bucket + the four packed forward weights + their transposed dgrad copies); every
        decoder layer's weight tensors are re-pointed at slot l % 2 and the model's own layer allocation is freed."""
        w, st, lay, dev = self.w, self.store, self.store.layout, self.device
        by_group = {id(u.group): u for u in st.units if u.group is not None}
        self._layer_units: Dict[Tuple[int, str], Unit] = {
            (l, key): by_group[id(w.groups[gi])] for l, key, gi, _ in w._layer_views if id(w.groups[gi]) in by_group}
        self._sharded_layers = sorted({l for l, _ in self._layer_units})
        sizes = {lay.buckets[self._layer_units[(l, "qkv_w")].bucket].numel for l in self._sharded_layers}
        assert len(sizes) == 1 and len(self._layer_units) == 4 * len(self._sharded_layers), "decoder layers are uniform units"
        bucket_numel = next(iter(sizes))
        L0 = w.layers[0]
        self._slots: List[dict] = []
        for _ in range(2):
            slot = {"flat": torch.zeros(bucket_numel, dtype=torch.bfloat16, device=dev)}
            for key in self._LAYER_KEYS:
                p = getattr(L0, key)
                slot[key] = torch.zeros(tuple(p.shape), dtype=torch.bfloat16, device=dev)
                if not self.fp8:
                    slot[key + "T"] = torch.zeros(p.shape[1] * 2, p.shape[0] // 2, 64, 8, dtype=torch.bfloat16, device=dev)
            slot["ready"], slot["free"] = torch.cuda.Event(), torch.cuda.Event()
            slot["grads_flushed"] = torch.cuda.Event()     # the gradient slot of the same parity has been reduced and kept
            self._slots.append(slot)
        self._slot_key = {slot[k].data_ptr(): k for slot in self._slots for k in self._LAYER_KEYS}
        if self.fp8:
            self._fp8_scratch()
            for slot in self._slots:
                for k in self._LAYER_KEYS:
                    p = slot[k]
                    self._w8[p.data_ptr()] = self._fp8_entry(p.shape[0] * 16, p.shape[1] * 32, dev)
        for slot in self._slots:
            for k in self._LAYER_KEYS:
                if k + "T" in slot:
                    self._wT[slot[k].data_ptr()] = slot[k + "T"]
        if len(self._sharded_layers) != self.dims.llm_layers:
            raise ValueError("parameter sharding needs every decoder layer trainable (vla-full-train / vla-train)")
        if w.layers_resident:
            w.release_layer_weights(self._slots)
        else:                                  # re-planned step over an already sharded model
            w.repoint_layer_weights(self._slots)
        # per-layer prepared pack ops: logical [n, k] rows of the gathered bucket → forward and dgrad layouts of the slot
        self._pack_ops: Dict[int, Tuple[List[Op], List[Op]]] = {}          # layer → (forward pass, backward pass)
        for l in self._sharded_layers:
            slot = self._slots[l % 2]
            b = lay.buckets[self._layer_units[(l, "qkv_w")].bucket]
            fwd, bwd = [], []
            for key in self._LAYER_KEYS:
                u = self._layer_units[(l, key)]
                rm = slot["flat"][u.offset - b.offset:u.offset - b.offset + u.numel].view(u.group.n, u.group.k)
                if self.fp8:                   # the slot carries the e4m3 copies (+ scales) instead of the bf16 layouts
                    q = self._fp8_weight_ops(rm, self._w8[slot[key].data_ptr()])
                    fwd += q[:2]
                    bwd += q[2:]
                    continue
                fwd.append(T.pack(rm, slot[key], run=False))                              # forward operand layout
                bwd.append(T.transpose_pack(rm, slot[key + "T"], u.group.n, run=False))   # dgrad operand layout
            self._pack_ops[l] = (fwd, bwd + fwd if self.recompute else bwd)


    # ---- the other FSDP units: ViT blocks + patch embeddings ("vision" pool), projector / token embeddings / lm_head ("head") ----
    def _setup_unit_pools(self) -> None:
        """Same machinery as the decoder layers for the reference's remaining FSDP units (prismatic.py:285-306,
        dinosiglip_vit.py:136-140, fsdp.py:160-168): per pool TWO gather slots — the logical bf16 bucket + one buffer of
        forward layouts + one of dgrad layouts, each sized for the pool's largest unit and carved into per-unit views — the
        model's tensors are re-pointed at the views of slot (unit index % 2) and the pool's own allocation is freed. A unit is
        all-gathered (and packed) into its slot one unit ahead of its use, on the communication stream; its weight gradients
        go to one of two transient fp32 slots of the pool and are reduce-scattered into the rank's persistent slice right
        after the unit's last weight-gradient GEMM. The token embeddings are a plain [vocab, D] tensor: read in place from the
        gathered bucket."""
        w, st, lay, dev = self.w, self.store, self.store.layout, self.device
        by_group = {id(u.group): u for u in st.units if u.group is not None}
        by_key = {b.key: b for b in lay.buckets}
        self._pool_slots: Dict[str, List[dict]] = {}
        for pool in ("vision", "head"):
            fields = w.unit_fields(pool)                                   # (unit key, path, group index | None, HF name | None)
            keys = list(dict.fromkeys(k for k, *_ in fields))
            live = [k for k in keys if k in st.sharded_keys]
            if not live:
                continue
            if len(live) != len(keys):
                raise ValueError(f"parameter sharding needs every unit of the {pool} pool trainable or none ({set(keys) - set(live)} frozen)")
            numel = lambda k: sum(w.groups[gi].n * w.groups[gi].k for kk, _, gi, _ in fields if kk == k and gi is not None)
            flat_max, pk_max = max(by_key[k].numel for k in keys), max(max(numel(k) for k in keys), 8)
            slots = [dict(flat=torch.zeros(flat_max, dtype=torch.bfloat16, device=dev),
                          fwd=torch.zeros(pk_max, dtype=torch.bfloat16, device=dev),
                          bwd=torch.zeros(pk_max, dtype=torch.bfloat16, device=dev),
                          ready=torch.cuda.Event(), free=torch.cuda.Event(), grads_flushed=torch.cuda.Event()) for _ in range(2)]
            self._pool_slots[pool] = slots
            views: Dict[tuple, torch.Tensor] = {}
            for k in keys:
                b = by_key[k]
                parity = st.grad_slot_of[k][1]
                slot, off = slots[parity], 0
                info = dict(pool=pool, parity=parity, bucket=b, by_ptr={}, T={}, fwd_ops=[], bwd_ops=[], plain=[])
                for kk, path, gi, name in fields:
                    if kk != k:
                        continue
                    if gi is not None:
                        u = by_group[id(w.groups[gi])]
                        n, kd = u.group.n, u.group.k
                        fv = slot["fwd"][off:off + n * kd].view(n // 16, kd // 32, 64, 8)
                        tv = slot["bwd"][off:off + n * kd].view(kd // 16, n // 32, 64, 8)
                        off += n * kd
                        rm = slot["flat"][u.offset - b.offset:u.offset - b.offset + u.numel].view(n, kd)
                        info["fwd_ops"].append(T.pack(rm, fv, run=False))
                        info["bwd_ops"].append(T.transpose_pack(rm, tv, n, run=False))
                        info["by_ptr"][fv.data_ptr()], info["T"][fv.data_ptr()] = u, tv
                        views[(k, path)] = fv
                    else:                                                  # token embeddings: used where the gather lands them
                        u = st.by_name[name]
                        obj, attr = w._attr_of(path)
                        views[(k, path)] = slot["flat"][u.offset - b.offset:u.offset - b.offset + u.numel].view(getattr(obj, attr).shape)
                        info["plain"].append((path, u))
                        u.dst = None                                       # the optimizer writes the rank's slice, nothing is copied back
                self._units[k] = info
            if w.pool_resident(pool):
                w.release_unit_weights(pool, views)
            else:                                  # re-planned step over an already sharded model
                w.repoint_unit_weights(pool, views)

    def _u_slot(self, key: str) -> dict:
        info = self._units[key]
        return self._pool_slots[info["pool"]][info["parity"]]

    def _u_gather(self, key: str, backward: bool = False, both: bool = False) -> Op:
        """Issue unit `key`'s parameter gather + packing (forward layouts, dgrad layouts, or both) on the communication stream."""
        def fn():
            info, slot, side = self._units[key], self._u_slot(key), self._comm_stream
            side.wait_event(slot["free"])                     # the unit that used this slot before is done with it
            with torch.cuda.stream(side):
                b = info["bucket"]
                self.comm.all_gather_into(slot["flat"][:b.numel], self.store.own_slice(b))
                if both or not backward:
                    ops.run_all(info["fwd_ops"])
                if both or backward:
                    ops.run_all(info["bwd_ops"])
                slot["ready"].record(side)
        return ops.glue("gather_unit_params", fn, ())

    def _u_await(self, key: str) -> Op:
        return ops.glue("await_unit_params", lambda: torch.cuda.current_stream().wait_event(self._u_slot(key)["ready"]), ())

    def _u_release(self, key: str) -> Op:
        return ops.glue("release_unit_params", lambda: self._u_slot(key)["free"].record(torch.cuda.current_stream()), ())

    def _u_await_grad(self, key: str) -> Op:
        return ops.glue("await_unit_grad_slot", lambda: torch.cuda.current_stream().wait_event(self._u_slot(key)["grads_flushed"]), ())

    def _u_flush(self, key: str) -> Op:
        """After the unit's last weight gradient: reduce-scatter its transient fp32 slot, keep this rank's slice."""
        def fn():
            st, lay, side = self.store, self.store.layout, self._comm_stream
            if not st.use_grad_slots:
                return
            b = self._units[key]["bucket"]
            gs = st.grad_slot(b)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                scratch = self._rs_scratch[:b.numel] if self._rs_scratch is not None else None
                self.comm.reduce_scatter_bucket(gs, b, scratch)
                n = lay.shard_numel(b)
                T.copy_f32(gs[lay.rank * n:(lay.rank + 1) * n], st.reduced_grad(b))
                self._u_slot(key)["grads_flushed"].record(side)
        return ops.glue("flush_unit_grads", fn, ())

    def _u_seq(self, seq: List[str], j: int, backward: bool, body: List[Op], gather: bool = True, flush: bool = True) -> List[Op]:
        """Ops of unit seq[j] wrapped in its parameter / gradient-slot protocol: wait for its gather, start the next unit's
        gather (after this unit's release when both share a slot), and in the backward pass wait for the gradient slot and
        flush it afterwards. Units that are not parameter-sharded pass through."""
        key = seq[j]
        if key not in self._units:
            return body
        nxt = seq[j + 1] if j + 1 < len(seq) and seq[j + 1] in self._units else None
        same = nxt is not None and self._u_slot(nxt) is self._u_slot(key)
        pre: List[Op] = []
        if j == 0 and gather:
            pre.append(self._u_gather(key, backward))
        pre.append(self._u_await(key))
        if nxt is not None and not same:
            pre.append(self._u_gather(nxt, backward))
        if backward:
            pre.append(self._u_await_grad(key))
        post = [self._u_release(key)]
        if backward and flush:
            post.append(self._u_flush(key))
        if nxt is not None and same:
            post.append(self._u_gather(nxt, backward))
        return pre + body + post

    def _layer_bucket(self, l: int):
        return self.store.layout.buckets[self._layer_units[(l, "qkv_w")].bucket]

    def _gather_ops(self, l: int, backward: bool = False) -> Op:
        """Issue layer l's parameter gather + packing on the communication stream (runs ahead of the compute stream)."""
        def fn():
            slot, side = self._slots[l % 2], self._comm_stream
            side.wait_event(slot["free"])                     # the layer that used this slot before is done with it
            with torch.cuda.stream(side):
                b = self._layer_bucket(l)
                self.comm.all_gather_into(slot["flat"][:b.numel], self.store.own_slice(b))
                ops.run_all(self._pack_ops[l][1 if backward else 0])
                slot["ready"].record(side)
        return ops.glue("gather_layer_params", fn, ())

    def _await_ops(self, l: int) -> Op:
        return ops.glue("await_layer_params", lambda: torch.cuda.current_stream().wait_event(self._slots[l % 2]["ready"]), ())

    def _release_ops(self, l: int) -> Op:
        return ops.glue("release_layer_params", lambda: self._slots[l % 2]["free"].record(torch.cuda.current_stream()), ())

    def _await_grad_slot_ops(self, l: int) -> Op:
        """Before layer l's weight-gradient GEMMs write gradient slot l % 2: the flush of the layer that used it last
        (l + 2) must be over."""
        return ops.glue("await_grad_slot", lambda: torch.cuda.current_stream().wait_event(self._slots[l % 2]["grads_flushed"]), ())

    def _flush_grads_ops(self, l: int) -> Op:
        """After layer l's last weight-gradient GEMM: on the communication stream, reduce-scatter the layer's full fp32
        gradient (slot l % 2) over the ranks — in place, this rank's slice of the slot receives the sum (bf16 wire copy
        when reduce_dtype says so) — and keep that slice in the persistent sharded gradient; the slot is then free for
        layer l - 2. The full gradient of a decoder layer therefore lives for two layers' worth of backward, as under
        FSDP (fsdp.py:160-168), instead of the whole step."""
        def fn():
            st, lay, side = self.store, self.store.layout, self._comm_stream
            if not st.use_grad_slots:                # nothing to reduce or move: the GEMMs wrote the persistent buffer
                return
            b = self._layer_bucket(l)
            slot = st.grad_slot(b)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                scratch = self._rs_scratch[:b.numel] if self._rs_scratch is not None else None
                self.comm.reduce_scatter_bucket(slot, b, scratch)
                n = lay.shard_numel(b)
                T.copy_f32(slot[lay.rank * n:(lay.rank + 1) * n], st.reduced_grad(b))
                self._slots[l % 2]["grads_flushed"].record(side)
        return ops.glue("flush_layer_grads", fn, ())

    def _unit_of(self, packed: torch.Tensor) -> Optional[Unit]:
        if self._cur_unit is not None and packed.data_ptr() in self._units[self._cur_unit]["by_ptr"]:
            return self._units[self._cur_unit]["by_ptr"][packed.data_ptr()]     # slot views are shared by the units of a pool
        if self.shard_params and packed.data_ptr() in self._slot_key:
            assert self._cur_layer >= 0, "slot weights are only addressed while a decoder layer is being planned"
            return self._layer_units[(self._cur_layer, self._slot_key[packed.data_ptr()])]
        return self.store.unit_of_packed(packed)

    def materialize_params(self) -> None:
        """Bring the decoder layers back into the model's own allocation (gather every layer, pack) — for inference,
        `state_dict()` / `save_pretrained` after parameter-sharded training. The step's plans address the slots, so this
        step object cannot train afterwards."""
        if not self.shard_params or self._materialized:
            return
        torch.cuda.synchronize()
        w, st = self.w, self.store
        w.restore_layer_weights()
        slot = self._slots[0]
        for l in self._sharded_layers:
            b = self._layer_bucket(l)
            self.comm.all_gather_into(slot["flat"][:b.numel], st.own_slice(b))
            for key in self._LAYER_KEYS:
                u = self._layer_units[(l, key)]
                rm = slot["flat"][u.offset - b.offset:u.offset - b.offset + u.numel].view(u.group.n, u.group.k)
                T.pack(rm, getattr(w.layers[l], key))
        for pool, slots in getattr(self, "_pool_slots", {}).items():
            w.restore_unit_weights(pool)
            by_group = {id(u.group): u for u in st.units if u.group is not None}
            for key, path, gi, name in w.unit_fields(pool):
                info = self._units[key]
                b, slot = info["bucket"], slots[info["parity"]]
                torch.cuda.synchronize()                    # a slot is re-used by the next unit of its parity
                self.comm.all_gather_into(slot["flat"][:b.numel], st.own_slice(b))
                obj, attr = w._attr_of(path)
                u = by_group[id(w.groups[gi])] if gi is not None else st.by_name[name]
                rm = slot["flat"][u.offset - b.offset:u.offset - b.offset + u.numel]
                if gi is not None:
                    T.pack(rm.view(u.group.n, u.group.k), getattr(obj, attr))
                else:
                    getattr(obj, attr).view(-1).copy_(rm)
        torch.cuda.synchronize()
        self._materialized = True

    def _g(self, *a, **k) -> Op:
        """Prepared GEMM with the split-K scratch: training has no batch-slot invariance to keep (engine.py keeps it
        off for inference), so ragged last rounds of K >= 8192 GEMMs are split along K."""
        return ops.gemm(*a, workspace=self.ws, run=False, **k)

    def wT(self, packed: torch.Tensor) -> torch.Tensor:
        """[K, N]-packed transposed copy of a packed [N, K] weight (built on first use, refreshed by repack)."""
        key = packed.data_ptr()
        if self._cur_unit is not None and key in self._units[self._cur_unit]["T"]:
            return self._units[self._cur_unit]["T"][key]
        assert not any(key in info["T"] for info in self._units.values()), "slot weights are addressed while their unit is planned"
        if key not in self._wT:
            n, k = packed.shape[0] * 16, packed.shape[1] * 32
            self._wT[key] = transposed_pack(packed, n, k)
        return self._wT[key]

    def _dgrad(self, dy: torch.Tensor, packed: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE, **kw) -> Op:
        return self._g(dy, self.wT(packed), out, epilogue, **kw)

    def _wgrad_into(self, dy: torch.Tensor, x: torch.Tensor, gview: torch.Tensor, fp8: bool = False) -> List[Op]:
        """gview[N, K] (fp32) = dyᵀ[N, T] · x[T, K]: the TN GEMM reads dy and x where they lie (transposing LDS reads);
        BL_WGRAD_NT=1 keeps the round-1 form — the NT GEMM over token-padded transposed copies — as the A/B reference.
        fp8=True: the e4m3 form (TrainStep(fp8_wgrad=True)) — transpose, quantise per channel, NT GEMM on bl_gemm_fp8."""
        Tn, N = dy.shape
        K = x.shape[1]
        assert tuple(gview.shape) == (N, K), (dy.shape, x.shape, gview.shape)
        if fp8:
            Tq = self._Tq
            qA = self._wg_qA[:N * Tq].view(N, Tq)
            pB = self._wg_pB[:K * Tq].view(torch.bfloat16).view(K // 16, Tq // 64, 64, 8)
            sA, sB = self._wg_sA[:N], self._wg_sB[:K]
            if WGRAD_FP8_5PASS:
                tA, tB = self._wg_tA[:N * Tq].view(N, Tq), self._wg_tB[:K * Tq].view(K, Tq)
                qB = self._wg_qB[:K * Tq].view(K, Tq)
                return [T.transpose_pad(dy, tA, Tq, run=False), ops.quantize_rows_fp8(tA, qA, sA, run=False)[2],
                        T.transpose_pad(x, tB, Tq, run=False), ops.quantize_rows_fp8(tB, qB, sB, run=False)[2],
                        T.pack(qB.view(torch.bfloat16), pB, run=False), ops.gemm_fp8(qA, sA, pB, sB, gview, EPI_F32, run=False)]
            # two passes per operand: column |max| (2 B read), then one transposing quantise(-and-pack) (2 B read, 1 B written)
            am = self._wg_amax[:N + K]
            return [T.fill_zero(am, run=False), T.colamax(dy, am[:N], run=False), T.colamax(x, am[N:], run=False),
                    T.transpose_quantize_fp8(dy, am[:N], self._wg_qA[:N * Tq], sA, Tq, False, run=False),
                    T.transpose_quantize_fp8(x, am[N:], self._wg_pB[:K * Tq], sB, Tq, True, run=False),
                    ops.gemm_fp8(qA, sA, pB, sB, gview, EPI_F32, run=False)]
        if not WGRAD_NT and N % 8 == 0 and K % 8 == 0:
            return [T.gemm_tn(dy, x, gview, workspace=self.ws, run=False)]
        Tp = (Tn + 63) // 64 * 64
        assert max(N, K) * Tp <= self.tA.numel()
        tA = self.tA[:N * Tp].view(N, Tp)
        tB = self.tB[:K * Tp].view(K, Tp)
        tBp = self.tBp[:K * Tp].view(K // 16, Tp // 32, 64, 8)
        if K % 64 == 0:                        # one pass: transpose straight into the packed operand layout
            prep = [T.transpose_pack(x, tBp, Tp, run=False)]
        else:
            prep = [T.transpose_pad(x, tB, Tp, run=False), T.pack(tB, tBp, run=False)]
        return [T.transpose_pad(dy, tA, Tp, run=False)] + prep + [self._g(tA, tBp, gview, EPI_F32, algo_nk=(K, Tn))]

    def _wgrad(self, dy: torch.Tensor, x: torch.Tensor, packed: torch.Tensor) -> List[Op]:
        """Weight gradient of a base linear; [] when the weight is frozen."""
        u = self._unit_of(packed)
        if u is None:
            return []
        want = self.fp8_wgrad and packed.data_ptr() in self._w8
        fp8 = want and dy.shape[1] % 16 == 0 and x.shape[1] % 16 == 0
        if want:                               # how many weight-gradient GEMMs really run on the e4m3 path (bench.py reports it)
            self.fp8_wgrad_gemms[0 if fp8 else 1] += 1
            if not fp8 and self.fp8_wgrad_gemms[1] == 1:
                import warnings
                warnings.warn(f"fp8_wgrad: a weight gradient of shape [{dy.shape[1]}, {x.shape[1]}] is not a multiple of 16 in both "
                              "dimensions and stays on the bf16 TN GEMM")
        return self._wgrad_into(dy, x, self.store.grad_view(u), fp8=fp8)

    def _build_extended_weights(self) -> None:
        """LoRA: K-concatenated packed weights [W | s·B] (forward) and [Wᵀ | Aᵀ] (dgrad) per adapted linear. The frozen
        part is copied once; the adapter columns are (re)written by `_adapter_ops` after every optimizer step."""
        lora, s = self.lora, self.lora.scaling
        dev = self.device
        zb = lambda *shape: torch.zeros(*shape, dtype=torch.bfloat16, device=dev)
        entries = []
        self._ATp: Dict[int, torch.Tensor] = {}                             # packed Aᵀ [k, R]: the u = dt·A operand (lora_dropout > 0)
        if lora.dropout > 0.0:
            rows = max([self.T, self.B * 256] + [self.B * tw.dims.tokens for tw in (self.w.dino, self.w.siglip)])
            kmax = max(ad.group.k for ad in lora.adapters)
            self._xd, self._ud = zb(rows, kmax), zb(rows, kmax)            # dropout(x) and u = dt·A of ONE adapted linear at a time
            self._drop_seed = torch.zeros(1, dtype=torch.int32, device=dev)     # bumped once per forward pass
        for ad in lora.adapters:
            packed, n, k, R = ad.group.packed, ad.group.n, ad.group.k, ad.R
            KT, NT = (k + R) // 32, (n + R) // 32
            We, WTe = zb(n // 16, KT, 64, 8), zb(k // 16, NT, 64, 8)
            We.view(n // 16, KT, 512)[:, :k // 32].copy_(packed.view(n // 16, k // 32, 512))
            WTe.view(k // 16, NT, 512)[:, :n // 32].copy_(transposed_pack(packed, n, k).view(k // 16, n // 32, 512))
            BsT = zb(R // 16, n // 32, 64, 8)
            key = packed.data_ptr()
            self._Wext[key], self._WText[key], self._BsT[key] = We, WTe, BsT
            # the adapter columns of the extended weights, from the live adapter tensors: [W | s·B], (s·B)ᵀ, A, [Wᵀ | Aᵀ] — table
            # entries of ONE bl_batched_ops launch for all adapters (rounds 1-3: five launches per adapter, 1 650 per step)
            entries += [T.be_pack(ad.B, We, KT, k // 32, scale=s), T.be_transpose_pack(ad.B, BsT, n, scale=s),
                        T.be_pack(ad.A, ad.A_p), T.be_transpose_pack(ad.A, WTe, R, NT, n // 32)]
            if lora.dropout > 0.0:
                self._ATp[key] = zb(k // 16, R // 32, 64, 8)
                entries.append(T.be_transpose_pack(ad.A, self._ATp[key], R))
        self._adapter_ops = [T.batched(entries, dev, run=False)]
        ops.run_all(self._adapter_ops)

    def _lin(self, x: torch.Tensor, packed: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE, **kw) -> List[Op]:
        """Forward of one nn.Linear. With an adapter: t = x·Aᵀ into the spare columns of x's buffer, then ONE GEMM
        y = [x | t]·[W | s·B]ᵀ over K + R."""
        ad = self.lora.get(packed) if self.lora is not None else None
        e8 = self._w8.get(packed.data_ptr())
        if e8 is not None:                                 # decoder-layer linear on the e4m3 MFMA path
            q = self._q8[x.shape[1]]
            return [ops.quantize_rows_fp8(x, q, self._sx, run=False)[2],
                    ops.gemm_fp8(q, self._sx, e8["w8"], e8["sw"], out, epilogue, run=False, **kw)]
        if ad is None:
            return [self._g(x, packed, out, epilogue, **kw)]
        K, R = x.shape[1], ad.R
        wide = self._wide[x.data_ptr()]
        assert wide.shape[1] >= K + R, "input buffer of an adapted linear lacks the adapter columns"
        t = wide[:, K:K + R]
        self._lora_t[packed.data_ptr()] = t
        pre: List[Op] = []
        xa = x
        if self.lora.dropout > 0.0:            # PEFT: lora_A(dropout(x)); the base product below still reads the un-masked x
            xa = self._xd[:x.shape[0], :K]
            pre = [T.dropout(x, xa, self.lora.dropout, self._drop_seed, ad.index, run=False)]
        return pre + [self._g(xa, ad.A_p, t, EPI_NONE), self._g(wide[:, :K + R], self._Wext[packed.data_ptr()], out, epilogue, **kw)]

    def _lin_bwd(self, dy: torch.Tensor, x: torch.Tensor, packed: torch.Tensor, dx: Optional[torch.Tensor],
                 epilogue: int = EPI_NONE, **kw) -> List[Op]:
        """Backward of one nn.Linear given dy: base wgrad (if trainable) and dx = dy·W (if wanted); with an adapter:
        dt = dy·(sB) into the spare columns of dy's buffer, dB = (s·tᵀ·dy)ᵀ and dA = dtᵀ·x through the small-output TN
        GEMM (dy / x read once, untransposed), and dx = [dy | dt]·[Wᵀ | Aᵀ]ᵀ as ONE GEMM."""
        ad = self.lora.get(packed) if self.lora is not None else None
        if ad is None:
            plan = self._wgrad(dy, x, packed)
            e8 = self._w8.get(packed.data_ptr())
            if dx is not None and e8 is not None:          # dx = dy · W on the e4m3 path (dy per token row, Wᵀ per input channel)
                q = self._q8_dy[:, :dy.shape[1]]
                plan += [ops.quantize_rows_fp8(dy, q, self._sdy, run=False)[2],
                         ops.gemm_fp8(q, self._sdy, e8["wT8"], e8["swT"], dx, EPI_NONE, run=False)]
            elif dx is not None:
                plan.append(self._dgrad(dy, packed, dx, epilogue, **kw))
            return plan
        i, st, s, R, N = ad.index, self.store, self.lora.scaling, ad.R, dy.shape[1]
        key = packed.data_ptr()
        wide = self._wide[dy.data_ptr()]
        assert wide.shape[1] >= N + R, "gradient buffer of an adapted linear lacks the adapter columns"
        dt = wide[:, N:N + R]
        t = self._lora_t[key]
        gB = st.grad_view(f"lora.{i}.B").view(ad.group.n, R)
        gA = st.grad_view(f"lora.{i}.A").view(R, ad.group.k)
        plan = [self._g(dy, self._BsT[key], dt, EPI_NONE), T.gemm_tn_small(t, dy, gB, True, self.tn_ws, alpha=s, run=False)]
        if len(ad.modules) > 1:
            plan.append(T.lora_block_mask(gB, R // len(ad.modules), len(ad.modules), ad.mode == "interleave", run=False))
        p = self.lora.dropout
        xa = x
        if p > 0.0:                            # the mask of the forward pass, recomputed (same device seed, same salt)
            xa = self._xd[:x.shape[0], :x.shape[1]]
            plan.append(T.dropout(x, xa, p, self._drop_seed, ad.index, run=False))
        plan.append(T.gemm_tn_small(dt, xa, gA, False, self.tn_ws, run=False))
        if dx is not None:
            plan.append(self._g(wide[:, :N + R], self._WText[key], dx, epilogue, **kw))
            if p > 0.0:                        # dx = dy·W + dt·A so far; dropout's backward masks the adapter's share
                assert epilogue == EPI_NONE, "activation-backward epilogues are not combined with lora_dropout"
                u = self._ud[:x.shape[0], :x.shape[1]]
                plan += [self._g(dt, self._ATp[key], u, EPI_NONE), T.dropout_grad_fix(u, dx, p, self._drop_seed, ad.index, run=False)]
        return plan

    def _lin_act(self, x, packed, pre, act, kind: str, **kw) -> List[Op]:
        """Linear + activation of the training forward with the PRE-activation kept (autograd's saved tensor): one GEMM
        whose epilogue writes both (`kind` "swiglu": gate/up interleaved → silu(g)·u; "gelu": bias + exact-erf GELU). The
        e4m3 path and BL_TRAIN_FUSED_ACT=0 (see above) keep the plain epilogue + an elementwise pass; both forms
        give the same bits."""
        fused = not UNFUSED_FWD and self._w8.get(packed.data_ptr()) is None
        if kind == "swiglu":
            if fused:
                return self._lin(x, packed, pre, EPI_SWIGLU_KEEP, out2=act, **kw)
            return self._lin(x, packed, pre, EPI_NONE, **kw) + [T.swiglu(pre, act, run=False)]
        if fused:
            return self._lin(x, packed, pre, EPI_BIAS_GELU_KEEP, out2=act, **kw)
        return self._lin(x, packed, pre, EPI_BIAS, **kw) + [T.gelu(pre, act, run=False)]

    def _lin_bwd_act(self, dy, x, packed, pre, dpre, dact, kind: str) -> List[Op]:
        """Backward of `act(pre)` → Linear(packed) given dy of the linear: the input-gradient GEMM's epilogue applies the
        activation's backward with the saved pre-activation, so d act never travels through HBM (dact is only used by the
        unfused forms: e4m3 dgrad and BL_TRAIN_FUSED_ACT=0)."""
        fused = not UNFUSED_BWD and self._w8.get(packed.data_ptr()) is None
        if self.lora is not None and self.lora.dropout > 0.0 and self.lora.get(packed) is not None:
            fused = False       # dropout's backward adds the masked adapter share to d act BEFORE the activation's backward
        if not fused:
            bwd = T.swiglu_backward(pre, dact, dpre, run=False) if kind == "swiglu" else T.gelu_backward(pre, dact, dpre, run=False)
            return self._lin_bwd(dy, x, packed, dact) + [bwd]
        return self._lin_bwd(dy, x, packed, dpre, epilogue=EPI_SWIGLU_BWD if kind == "swiglu" else EPI_GELU_BWD, res=pre)

    def _gvec(self, name: str, n: int) -> torch.Tensor:
        """fp32 gradient slot of a vector parameter, or a scratch sink when it is frozen."""
        return self.store.grad_view(name) if self.store.trainable(name) else self._frozen_dw[:n]

    def _bias_grad(self, dy: torch.Tensor, name: str) -> List[Op]:
        """Bias gradient = column sums of dy; nothing when the bias is frozen."""
        return [T.colsum(dy, self.store.grad_view(name), self.col_ws, run=False)] if self.store.trainable(name) else []

    # ---- plans --------------------------------------------------------------------------------------------------
    def _plan_forward(self) -> List[Op]:
        d, w, B, S = self.dims, self.w, self.B, self.S
        D, H, hd = d.llm_dim, d.llm_heads, d.head_dim
        g = self._g
        plan: List[Op] = []
        head = "projector" in self._units          # the head pool is parameter-sharded: projector, token embeddings, lm_head
        if head:                                   # lm_head keeps its slot (both layouts) from here to its dgrad in the backward pass
            plan += [self._u_gather("projector"), self._u_gather("llm.lm_head", both=True), self._u_await("projector")]
        # projector with the pre-activations kept (modeling_prismatic.py:151-156)
        lin = self._lin
        plan += self._lin_act(self.feats, w.fc1_w, self.z1, self.p1, "gelu", bias=w.fc1_b)
        plan += self._lin_act(self.p1, w.fc2_w, self.z2, self.p2, "gelu", bias=w.fc2_b)
        plan += lin(self.p2, w.fc3_w, self.p3, EPI_BIAS, bias=w.fc3_b)
        if head:                                   # the embeddings share the projector's slot: gathered once it is released
            assert self._u_slot("llm.embed") is self._u_slot("projector") and self._u_slot("llm.lm_head") is not self._u_slot("projector")
            plan += [self._u_release("projector"), self._u_gather("llm.embed")]
        plan.append(T.map_rows(self.p3, self.x[0], rows=B * 256, group=256, stride=S, offset=1, scatter=True, run=False))
        if head:
            plan.append(self._u_await("llm.embed"))
        plan.append(ops.embed_splice(self.input_ids, w.embed, self.x[0].view(B, S, D), d.n_patches, run=False))
        if head:
            plan.append(self._u_release("llm.embed"))
        NL = d.llm_layers
        if self.shard_params:
            plan.append(self._gather_ops(0))
        for l in range(NL):
            if self.shard_params:          # this layer's weights have landed; the next layer's gather runs under its compute
                plan.append(self._await_ops(l))
                if l + 1 < NL:
                    plan.append(self._gather_ops(l + 1))
            plan += self._layer_forward(l)
            if self.shard_params:
                plan.append(self._release_ops(l))
        if head:
            plan.append(self._u_await("llm.lm_head"))
        plan += [ops.rmsnorm(self.x[-1], w.norm, self.hn, d.rms_eps, run=False),
                 g(self.hn, w.lm_head, self.logits, EPI_F32_BF16R),
                 ops.cross_entropy(self.logits, self.targets, self.row_loss, self.mean_cnt, IGNORE_INDEX, run=False)]
        return plan

    def _layer_forward(self, l: int, with_down: bool = True) -> List[Op]:
        """Forward ops of decoder layer l from its saved input x[l] (modeling_llama.py LlamaDecoderLayer). The backward
        plan replays them (without the down projection, whose output is x[l+1]) when activations are recomputed."""
        d, w, B, S = self.dims, self.w, self.B, self.S
        D, H, hd = d.llm_dim, d.llm_heads, d.head_dim
        lin, lw = self._lin, w.layers[l]
        lq, lo = self.qkv[0].stride(0), self.ao[0].stride(0)
        st, so = (S * lq, hd, lq), (S * lo, hd, lo)
        x, xm, qkv = self.x[l], self.xm[l], self.qkv[l]
        plan = [ops.rmsnorm(x, lw.ln1, self.h1[l], d.rms_eps, run=False)] + lin(self.h1[l], lw.qkv_w, qkv, EPI_NONE)
        plan += [T.rope(qkv, self.cos, self.sin, B=B, S=S, H=H, head_dim=hd, run=False),
                 T.attention_lse(qkv, qkv[:, D:], qkv[:, 2 * D:], self.ao[l], self.lse[l], B=B, H=H, Sq=S, Skv=S,
                                 head_dim=hd, q_strides=st, k_strides=st, v_strides=st, o_strides=so,
                                 causal=True, key_mask=self.key_mask, run=False)]
        plan += lin(self.ao[l], lw.o_w, xm, EPI_RES, res=x)
        plan += [ops.rmsnorm(xm, lw.ln2, self.h2[l], d.rms_eps, run=False)]
        plan += self._lin_act(self.h2[l], lw.gu_w, self.gu[l], self.act[l], "swiglu")
        if with_down:
            plan += lin(self.act[l], lw.down_w, self.x[l + 1], EPI_RES, res=xm)
        return plan

    def _plan_backward(self) -> List[Op]:
        d, w, B, S, st = self.dims, self.w, self.B, self.S, self.store
        D, H, hd = d.llm_dim, d.llm_heads, d.head_dim
        lm = "language_model.model"
        plan: List[Op] = [T.cross_entropy_backward(self.logits, self.targets, self.mean_cnt, self.dlogits, IGNORE_INDEX, run=False)]
        head = "projector" in self._units
        if head:
            plan.append(self._u_await_grad("llm.lm_head"))
            self._cur_unit = "llm.lm_head"
        plan += self._wgrad(self.dlogits, self.hn, w.lm_head)
        self._ready.append((len(plan), "llm.lm_head"))
        plan.append(self._dgrad(self.dlogits, w.lm_head, self.dh))
        self._cur_unit = None
        if head:                                   # the projector's dgrad layouts travel under the whole decoder backward
            plan += [self._u_release("llm.lm_head"), self._u_flush("llm.lm_head"), self._u_gather("projector", backward=True)]
        lb = self._lin_bwd
        dx, dx2 = self.dxa, self.dxb
        plan.append(T.rmsnorm_backward(self.x[-1], w.norm, self.dh, dx, self._gvec(f"{lm}.norm.weight", D), self.norm_ws,
                                       d.rms_eps, run=False))
        lq, lo = self.qkv[0].stride(0), self.ao[0].stride(0)
        assert self.dqkv.stride(0) == lq and self.dao.stride(0) == lo
        strides, so = (S * lq, hd, lq), (S * lo, hd, lo)
        stop_layer = self._lowest_needed_layer()
        if self.shard_params:
            plan.append(self._gather_ops(d.llm_layers - 1, backward=True))
        for l in range(d.llm_layers - 1, stop_layer - 1, -1):
            lw, b = w.layers[l], f"{lm}.layers.{l}"
            self._cur_layer = l
            if self.shard_params:
                plan.append(self._await_ops(l))
                if l - 1 >= stop_layer:
                    plan.append(self._gather_ops(l - 1, backward=True))
                plan.append(self._await_grad_slot_ops(l))
            if self.recompute:         # incl. the top layer: the plan stays idempotent (graph capture runs it twice)
                plan += self._layer_forward(l, with_down=False)
            plan += self._lin_bwd_act(dx, self.act[l], lw.down_w, self.gu[l], self.dgu, self.dact, "swiglu")
            plan += lb(self.dgu, self.h2[l], lw.gu_w, self.dh)
            plan.append(T.rmsnorm_backward(self.xm[l], lw.ln2, self.dh, dx2, self._gvec(f"{b}.post_attention_layernorm.weight", D),
                                           self.norm_ws, d.rms_eps, dres=dx, run=False))
            plan += lb(dx2, self.ao[l], lw.o_w, self.dao)
            qkv, dq = self.qkv[l], self.dqkv
            plan.append(T.attention_backward(qkv, qkv[:, D:], qkv[:, 2 * D:], self.ao[l], self.dao, self.lse[l], self.delta,
                                             dq, dq[:, D:], dq[:, 2 * D:], B=B, H=H, Sq=S, Skv=S, head_dim=hd,
                                             q_strides=strides, k_strides=strides, v_strides=strides,
                                             o_strides=so, causal=True, key_mask=self.key_mask, run=False))
            plan.append(T.rope_backward(dq, self.cos, self.sin, B=B, S=S, H=H, head_dim=hd, run=False))
            plan += lb(dq, self.h1[l], lw.qkv_w, self.dh)
            if self.shard_params:
                plan.append(self._release_ops(l))
                plan.append(self._flush_grads_ops(l))      # reduce-scatter + keep this rank's slice; frees the gradient slot
            self._ready.append((len(plan), f"llm.layer{l:02d}"))
            plan.append(T.rmsnorm_backward(self.x[l], lw.ln1, self.dh, dx, self._gvec(f"{b}.input_layernorm.weight", D),
                                           self.norm_ws, d.rms_eps, dres=dx2, run=False))
        self._cur_layer = -1
        if stop_layer > 0:
            return plan
        # dx = gradient of inputs_embeds [B, S, D]
        if st.trainable(f"{lm}.embed_tokens.weight"):
            if head:
                plan.append(self._u_await_grad("llm.embed"))
            plan.append(T.fill_zero(st.grad_view(f"{lm}.embed_tokens.weight"), run=False))     # accumulated with atomics
            plan.append(T.embed_backward(self.input_ids, dx.view(B, S, D), st.grad_view(f"{lm}.embed_tokens.weight").view(d.vocab, D),
                                         d.n_patches, run=False))
            if head:
                plan.append(self._u_flush("llm.embed"))
        if head:
            plan += [self._u_await("projector"), self._u_await_grad("projector")]
            self._cur_unit = "projector"
        if st.trainable("projector.fc3.weight") or self.lora is not None:
            plan.append(T.map_rows(dx, self.dp3, rows=B * 256, group=256, stride=S, offset=1, scatter=False, run=False))
            plan += self._bias_grad(self.dp3, "projector.fc3.bias")
            plan += self._lin_bwd_act(self.dp3, self.p2, w.fc3_w, self.z2, self.dz2, self.dp2, "gelu")
            plan += self._bias_grad(self.dz2, "projector.fc2.bias")
            plan += self._lin_bwd_act(self.dz2, self.p1, w.fc2_w, self.z1, self.dz1, self.dp1, "gelu")
            plan += self._bias_grad(self.dz1, "projector.fc1.bias")
            plan += lb(self.dz1, self.feats, w.fc1_w, self.dfeats if self.train_vision else None)
            self._ready.append((len(plan), "projector"))
            self._cur_unit = None
            if head:
                plan += [self._u_release("projector"), self._u_flush("projector")]
            if self.train_vision:
                self._bwd_tower_at = [len(plan)]                   # [start of tower 0, start of tower 1]: see _vis2
                for ti, (tw, sv, col) in enumerate(zip((w.dino, w.siglip), self.vis, (0, d.dino.dim))):
                    if ti:
                        self._bwd_tower_at.append(len(plan))
                    with self._side_scratch(ti == 1):
                        plan += self._plan_tower_backward(tw, col, sv, len(plan))
        return plan

    # ---- vision towers in training form (timm VisionTransformer blocks) ---------------------------
    def _alloc_tower(self, tw) -> dict:
        t, B = tw.dims, self.B
        M, Dm, Hp, n = B * t.tokens, t.dim, t.mlp_pad, t.n_run
        z = lambda *shape, dtype=torch.bfloat16: torch.zeros(*shape, dtype=dtype, device=self.device)
        pad = (t.tokens + 31) // 32 * 32
        ls = t.layerscale
        za, R_, b0 = self._za, self._R, tw.blocks[0]
        rr = max(R_(b0.fc2_w), R_(b0.proj_w))                  # dx / dx2 / du are the dy of fc2 and proj
        return dict(
            x=[z(M, Dm) for _ in range(n + 1)], h1=[za(M, Dm, R_(b0.qkv_w)) for _ in range(n)], qkv=[za(M, 3 * Dm, R_(b0.qkv_w)) for _ in range(n)],
            ao=[za(M, Dm, R_(b0.proj_w)) for _ in range(n)], lse=[z(B * t.heads * pad, dtype=torch.float32) for _ in range(n)],
            xm=[z(M, Dm) for _ in range(n)], h2=[za(M, Dm, R_(b0.fc1_w)) for _ in range(n)], zz=[z(M, Hp) for _ in range(n)],
            f=[za(M, Hp, R_(b0.fc2_w)) for _ in range(n)], u1=[z(M, Dm) if ls else None for _ in range(n)],
            u2=[z(M, Dm) if ls else None for _ in range(n)], delta=z(B * t.heads * pad, dtype=torch.float32),
            dxa=za(M, Dm, rr), dxb=za(M, Dm, rr), dh=z(M, Dm), du=za(M, Dm, rr), dao=za(M, Dm, R_(b0.proj_w)), dqkv=za(M, 3 * Dm, R_(b0.qkv_w)),
            df=z(M, Hp), dz=za(M, Hp, R_(b0.fc1_w)),
            dpe=z(B * 256, Dm), tok=z(max(t.n_prefix, 1) * Dm, dtype=torch.float32))

    def _plan_tower_forward(self, tw, feat_col: int, sv: dict) -> List[Op]:
        t, B, eps = tw.dims, self.B, self.dims.ln_eps
        Tk, Dm, hd = t.tokens, t.dim, t.head_dim
        col = self._vis.vbuf[1 if feat_col else 0]["col"]          # one im2col buffer per tower (kept for the patch wgrad)
        g = self._g
        x0 = sv["x"][0]
        seq, j0 = self._vseq_fwd, self._vseq_fwd.index(self._vkey(tw, -1))
        whole: List[Op] = []                       # the tower's plan; `plan` below collects one unit's ops at a time
        plan = [ops.im2col_patch14(self.pixel_values, t.chan0, col, run=False)]
        if tw.prefix is not None:
            plan.append(ops.write_prefix_tokens(tw.prefix, x0, B, Tk, run=False))
        plan.append(g(col, tw.patch_w, x0, ops.EPI_BIAS_RES, bias=tw.patch_b, res=tw.pos, res_row_mod=256,
                      out_map=(256, Tk, t.n_prefix)))
        whole += self._u_seq(seq, j0, False, plan)
        lq, lo = sv["qkv"][0].stride(0), sv["ao"][0].stride(0)
        st, so = (Tk * lq, hd, lq), (Tk * lo, hd, lo)
        for i, b in enumerate(tw.blocks):
            if i:
                whole += self._u_seq(seq, j0 + i, False, plan)
            x, xm, qkv = sv["x"][i], sv["xm"][i], sv["qkv"][i]
            plan = [ops.layernorm(x, b.norm1_w, b.norm1_b, sv["h1"][i], eps, run=False)]
            plan += self._lin(sv["h1"][i], b.qkv_w, qkv, EPI_BIAS, bias=b.qkv_b)
            plan += [T.attention_lse(qkv, qkv[:, Dm:], qkv[:, 2 * Dm:], sv["ao"][i], sv["lse"][i], B=B, H=t.heads, Sq=Tk,
                                     Skv=Tk, head_dim=hd, q_strides=st, k_strides=st, v_strides=st,
                                     o_strides=so, causal=False, run=False)]
            if b.ls1 is not None:
                plan += self._lin(sv["ao"][i], b.proj_w, sv["u1"][i], EPI_BIAS, bias=b.proj_b)
                plan.append(T.scale_residual(sv["u1"][i], b.ls1, x, xm, run=False))
            else:
                plan += self._lin(sv["ao"][i], b.proj_w, xm, ops.EPI_BIAS_RES, bias=b.proj_b, res=x)
            plan.append(ops.layernorm(xm, b.norm2_w, b.norm2_b, sv["h2"][i], eps, run=False))
            plan += self._lin_act(sv["h2"][i], b.fc1_w, sv["zz"][i], sv["f"][i], "gelu", bias=b.fc1_b)
            if b.ls2 is not None:
                plan += self._lin(sv["f"][i], b.fc2_w, sv["u2"][i], EPI_BIAS, bias=b.fc2_b)
                plan.append(T.scale_residual(sv["u2"][i], b.ls2, xm, sv["x"][i + 1], run=False))
            else:
                plan += self._lin(sv["f"][i], b.fc2_w, sv["x"][i + 1], ops.EPI_BIAS_RES, bias=b.fc2_b, res=xm)
        whole += self._u_seq(seq, j0 + len(tw.blocks), False, plan)
        # tap: patch rows of the last block output → this tower's channels of the fused feature map
        whole.append(T.map_rows(sv["x"][-1], self.feats[:, feat_col:feat_col + Dm], rows=B * 256, group=256, stride=Tk,
                                offset=t.n_prefix, scatter=False, run=False))
        return whole

    def _plan_tower_backward(self, tw, feat_col: int, sv: dict, base: int) -> List[Op]:
        """`base` = ops already in the backward plan (bucket-ready markers are absolute positions)."""
        t, B, eps, st = tw.dims, self.B, self.dims.ln_eps, self.store
        Tk, Dm, hd, M = t.tokens, t.dim, t.head_dim, B * t.tokens
        p = t.prefix
        tower_key = p.split(".")[1]
        gv = lambda name: self._gvec(name, Dm)
        gb = lambda name: st.grad_view(name) if st.trainable(name) else self._frozen_dw[Dm:2 * Dm]   # 2nd sink: LN dw + db
        lb = self._lin_bwd
        dx, dx2 = sv["dxa"], sv["dxb"]
        seq, j0 = self._vseq_bwd, self._vseq_bwd.index(self._vkey(tw, t.n_run - 1))
        whole: List[Op] = [T.fill_zero(self._wide.get(dx.data_ptr(), dx), run=False),
                           T.map_rows(self.dfeats[:, feat_col:feat_col + Dm], dx, rows=B * 256, group=256, stride=Tk,
                                      offset=t.n_prefix, scatter=True, run=False)]
        lq, lo = sv["qkv"][0].stride(0), sv["ao"][0].stride(0)
        assert sv["dqkv"].stride(0) == lq and sv["dao"].stride(0) == lo
        strides, so = (Tk * lq, hd, lq), (Tk * lo, hd, lo)
        for i in range(t.n_run - 1, -1, -1):
            b, bn = tw.blocks[i], f"{p}.blocks.{i}"
            dbr = dx
            plan: List[Op] = []                    # this block's ops: one parameter-sharded unit
            self._cur_unit = self._vkey(tw, i) if self._vkey(tw, i) in self._units else None
            if b.ls2 is not None:
                plan.append(T.layerscale_backward(dx, sv["u2"][i], b.ls2, sv["du"], gv(f"{bn}.ls2.scale_factor"), self.col_ws, run=False))
                dbr = sv["du"]
            plan += self._bias_grad(dbr, f"{bn}.mlp.fc2.bias")
            plan += self._lin_bwd_act(dbr, sv["f"][i], b.fc2_w, sv["zz"][i], sv["dz"], sv["df"], "gelu")
            plan += self._bias_grad(sv["dz"][:, :t.mlp], f"{bn}.mlp.fc1.bias")
            plan += lb(sv["dz"], sv["h2"][i], b.fc1_w, sv["dh"])
            plan.append(T.layernorm_backward(sv["xm"][i], b.norm2_w, sv["dh"], dx2, gv(f"{bn}.norm2.weight"), gb(f"{bn}.norm2.bias"),
                                             self.norm_ws, eps, dres=dx, run=False))
            dbr = dx2
            if b.ls1 is not None:
                plan.append(T.layerscale_backward(dx2, sv["u1"][i], b.ls1, sv["du"], gv(f"{bn}.ls1.scale_factor"), self.col_ws, run=False))
                dbr = sv["du"]
            plan += self._bias_grad(dbr, f"{bn}.attn.proj.bias")
            plan += lb(dbr, sv["ao"][i], b.proj_w, sv["dao"])
            qkv, dq = sv["qkv"][i], sv["dqkv"]
            plan.append(T.attention_backward(qkv, qkv[:, Dm:], qkv[:, 2 * Dm:], sv["ao"][i], sv["dao"], sv["lse"][i], sv["delta"],
                                             dq, dq[:, Dm:], dq[:, 2 * Dm:], B=B, H=t.heads, Sq=Tk, Skv=Tk, head_dim=hd,
                                             q_strides=strides, k_strides=strides, v_strides=strides,
                                             o_strides=so, causal=False, run=False))
            plan += self._bias_grad(dq, f"{bn}.attn.qkv.bias")
            plan += lb(dq, sv["h1"][i], b.qkv_w, sv["dh"])
            self._cur_unit = None
            whole += self._u_seq(seq, j0 + (t.n_run - 1 - i), True, plan)
            self._ready.append((base + len(whole), f"vision.{tower_key}.block{i:02d}"))
            whole.append(T.layernorm_backward(sv["x"][i], b.norm1_w, sv["dh"], dx, gv(f"{bn}.norm1.weight"), gb(f"{bn}.norm1.bias"),
                                              self.norm_ws, eps, dres=dx2, run=False))
        # stem: dx = gradient of [prefix tokens | patch embeddings + pos]
        plan = []
        self._cur_unit = self._vkey(tw, -1) if self._vkey(tw, -1) in self._units else None
        if st.trainable(f"{p}.pos_embed"):
            dxv = dx.view(B, Tk * Dm)
            plan.append(T.colsum(dxv[:, t.n_prefix * Dm:], st.grad_view(f"{p}.pos_embed"), self.col_ws, run=False))
            if t.n_prefix:
                plan.append(T.colsum(dxv[:, :t.n_prefix * Dm], sv["tok"], self.col_ws, run=False))
                plan.append(T.copy_f32(sv["tok"][:Dm], st.grad_view(f"{p}.cls_token"), run=False))
                plan.append(T.copy_f32(sv["tok"][Dm:t.n_prefix * Dm], st.grad_view(f"{p}.reg_token"), run=False))
            plan.append(T.map_rows(dx, sv["dpe"], rows=B * 256, group=256, stride=Tk, offset=t.n_prefix, scatter=False, run=False))
            plan += self._bias_grad(sv["dpe"], f"{p}.patch_embed.proj.bias")
            plan += self._wgrad(sv["dpe"], self._vis.vbuf[1 if feat_col else 0]["col"], tw.patch_w)
        self._cur_unit = None
        whole += self._u_seq(seq, j0 + t.n_run, True, plan)
        self._ready.append((base + len(whole), f"vision.{tower_key}.stem"))
        return whole

    def _lowest_needed_layer(self) -> int:
        """Backward stops above the lowest decoder layer that still has a trainable tensor below or inside it."""
        names = self.store.names
        if self.lora is not None:
            return 0
        if any(not n.startswith("language_model.model.layers.") and not n.startswith("language_model.lm_head")
               and not n.startswith("language_model.model.norm") for n in names):
            return 0                     # embeddings / projector / vision sit under layer 0
        lows = [int(n.split(".")[3]) for n in names if n.startswith("language_model.model.layers.")]
        return min(lows) if lows else self.dims.llm_layers

    def _plan_repack(self) -> List[Op]:
        """After AdamW wrote the bf16 copy of every group's logical matrix: refresh the forward and the dgrad layouts."""
        plan: List[Op] = []
        st = self.store
        nmax = max((u.numel for u in st.units), default=8)
        self._tW = torch.zeros(nmax, dtype=torch.bfloat16, device=self.device)
        # plain tensors (norm scales, biases, embeddings, LoRA adapters): updated bf16 values → the live tensors, as prepared
        # byte copies inside the plan (hundreds of them under LoRA: replayed with the graph instead of launched one by one)
        # ONE bl_batched_ops launch for the write-backs of the plain tensors and the re-packs of the GEMM weights (independent of
        # each other: all read the optimizer's bf16 staging copy) — ≈ 800 launches of the full fine-tune's plan, ≈ 700 of LoRA's
        entries = []
        for u in st.units:
            if u.group is None and st.layout.buckets[u.bucket].key not in st.sharded_keys:
                assert u.dst.is_contiguous()
                entries.append(T.be_copy(st.stage_view(u.offset, u.numel), u.dst))
        for u in st.units:
            if u.group is None or st.layout.buckets[u.bucket].key in st.sharded_keys:
                continue                                          # parameter-sharded layers are packed when gathered
            n, k = u.group.n, u.group.k
            rm = st.stage_view(u.offset, u.numel).view(n, k)
            key = u.group.packed.data_ptr()
            entries.append(T.be_pack(rm, u.group.packed))
            if key in self._wT:                                   # only weights that a dgrad GEMM actually reads
                if k % 64 == 0 and n % 32 == 0:
                    entries.append(T.be_transpose_pack(rm, self._wT[key], n))
                else:
                    plan.append(T.transpose_pack(rm, self._wT[key], n, run=False))
            if key in self._w8:                                   # e4m3 copies follow the updated weight
                plan += self._fp8_weight_ops(rm, self._w8[key])
        if entries:
            plan.insert(0, T.batched(entries, self.device, run=False))
        plan += self._adapter_ops                                 # LoRA: adapter columns of the K-concatenated weights
        return plan

    # ---- running ------------------------------------------------------------------------------------------------
    def set_batch(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], pixel_values: torch.Tensor,
                  labels: torch.Tensor) -> None:
        """Right-padded batch (PaddedCollatorForActionPrediction, data_utils.py:101-142) → static device inputs. Batches
        shorter than the planned L are padded further (pad id 32000 / mask 0 / label -100: no effect on valid rows)."""
        dev, B, L, P, S = self.device, self.B, self.L, self.dims.n_patches, self.S
        b, l = input_ids.shape
        if b != B or l > L:
            raise ValueError(f"batch {tuple(input_ids.shape)} does not fit the planned step ({B}, <= {L})")
        ids = torch.full((B, L), 32000, dtype=torch.int64, device=dev)
        ids[:, :l] = input_ids.to(dev)
        m = torch.zeros(B, L, dtype=torch.uint8, device=dev)
        m[:, :l] = (attention_mask.to(dev) if attention_mask is not None else torch.ones(b, l, device=dev)).to(torch.uint8)
        lab = torch.full((B, L), IGNORE_INDEX, dtype=torch.int64, device=dev)
        lab[:, :l] = labels.to(dev)
        self.input_ids.copy_(ids)
        self.pixel_values.copy_(pixel_values.to(dev).to(torch.bfloat16))
        self.key_mask[:, :1] = m[:, :1]
        self.key_mask[:, 1:1 + P] = 1
        self.key_mask[:, 1 + P:] = m[:, 1:]
        full = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64, device=dev)       # labels with 256 ignored patch columns
        full[:, :1] = lab[:, :1]
        full[:, 1 + P:] = lab[:, 1:]
        tg = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64, device=dev)
        tg[:, :-1] = full[:, 1:]                                                       # position t predicts token t+1
        self.targets.copy_(tg.view(-1))

    @contextlib.contextmanager
    def _side_scratch(self, on: bool):
        """While the SECOND tower's plans are built: the scratch buffers its ops bind are the side stream's own."""
        if not (on and self._vis2):
            yield
            return
        keep = {k: getattr(self, k) for k in self._scratch2}
        for k, v in self._scratch2.items():
            setattr(self, k, v)
        try:
            yield
        finally:
            for k, v in keep.items():
                setattr(self, k, v)

    def _run_forked(self, main_ops: List[Op], side_ops: List[Op]) -> None:
        """main_ops on the current stream, side_ops on the vision side stream, joined at the end (fork / join with stream
        waits, which HIP-graph capture records as parallel branches)."""
        main = torch.cuda.current_stream()
        self._vis_stream.wait_stream(main)
        with torch.cuda.stream(self._vis_stream):
            ops.run_all(side_ops)
        ops.run_all(main_ops)
        main.wait_stream(self._vis_stream)

    def _replay(self, key: str, plan, graph: bool) -> None:
        if self._materialized:
            raise RuntimeError("materialize_params() ended this step object's training (its plans address the gather slots)")
        run = plan if callable(plan) else (lambda: ops.run_all(plan))      # a callable issues its own (forked) launches
        if not graph or (self.shard_params and key in ("vfwd", "fwd", "bwd")):      # per-unit collectives stay out of HIP graphs
            run()
            return
        gr = self._graphs.get(key)
        if gr is None:
            run()                                                  # warm (lazy hipFuncSetAttribute etc.) outside capture
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                run()
            self._graphs[key] = gr
        gr.replay()

    def forward(self, graph: bool = False) -> torch.Tensor:
        """Vision towers (frozen) → projector → decoder → loss. Returns the device scalar loss."""
        if self.lora is not None and self.lora.dropout > 0.0:
            self._drop_seed.add_(1)            # a fresh mask per forward pass; the backward pass recomputes it from the same value
        if self.train_vision and self._vis2:
            self._replay("vfwd", lambda: self._run_forked(self._vfwd_tower[0], self._vfwd_tower[1]), graph)
        elif self.train_vision:
            self._replay("vfwd", self.vision_forward_ops, graph)
        else:
            self._vis.run_vision()
        self._replay("fwd", self.forward_ops, graph)
        return self.mean_cnt[0]

    def backward(self, graph: bool = False) -> None:
        """Hand-written backward into the flat fp32 gradient buffer. Sharded runs (world > 1) issue each bucket's
        reduce-scatter on a side stream as soon as its last wgrad is enqueued (overlapping the rest of the backward)."""
        st, lay = self.store, self.store.layout
        if not self.comm.active:
            if self._vis2 and getattr(self, "_bwd_tower_at", None) and len(self._bwd_tower_at) == 2:
                a, b = self._bwd_tower_at
                bo = self.backward_ops
                self._replay("bwd", lambda: (ops.run_all(bo[:a]), self._run_forked(bo[a:b], bo[b:])), graph)
            else:
                self._replay("bwd", self.backward_ops, graph)
            if self.shard_params:                                  # the per-layer gradient flushes ran on the side stream
                torch.cuda.current_stream().wait_stream(self._comm_stream)
            return
        self.mean_cnt[1].mul_(self.world)                          # dlogits / world: the SUM over ranks is the DDP mean
        main = torch.cuda.current_stream()
        by_key = {b.key: b for b in lay.buckets}
        done, reduced = 0, set()

        def reduce(b):
            reduced.add(b.key)
            if b.key in st.sharded_keys:                           # reduced by the flush step inside the plan
                return
            ev = torch.cuda.Event()
            ev.record(main)
            self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                self.comm.reduce_scatter_bucket(st.grad_range(b.offset, b.numel), b, st.stage_view(b.offset, b.numel))
        for upto, key in self._ready:
            ops.run_all(self.backward_ops[done:upto])
            done = upto
            b = by_key.get(key) or by_key.get("lora." + key)
            if b is not None:
                reduce(b)
        ops.run_all(self.backward_ops[done:])
        for i in comm_order(lay.buckets):                          # whatever has no marker (plain tensors) goes last
            if lay.buckets[i].key not in reduced:
                reduce(lay.buckets[i])
        main.wait_stream(self._comm_stream)

    def accumulate(self, scale: float = 1.0) -> None:
        """Gradient accumulation: add scale × (this micro-batch's gradients) to the accumulator (allocated on first use).
        `use_accumulated()` then makes the accumulated sum the gradient the optimizer step sees."""
        st, lay = self.store, self.store.layout
        if getattr(st, "grad_acc", None) is None:
            st.grad_acc = torch.zeros(max(lay.local_total, 8), dtype=torch.float32, device=self.device)
        # after backward() every bucket's gradient is reduced over the ranks and this rank's slice is in place: the
        # accumulator holds slices only (1/world of the model), so accumulation works unchanged under the sharded optimizer
        for b in lay.buckets:
            o, n = lay.local_offset(b), lay.shard_numel(b)
            T.axpy(st.grad_acc[o:o + n], st.reduced_grad(b), scale)

    def use_accumulated(self) -> None:
        st, lay = self.store, self.store.layout
        for b in lay.buckets:               # D2D copies (plumbing); the accumulator is cleared for the next window
            o, n = lay.local_offset(b), lay.shard_numel(b)
            T.copy_f32(st.grad_acc[o:o + n], st.reduced_grad(b))
        T.fill_zero(st.grad_acc)

    def clip_grad_norm(self) -> torch.Tensor:
        """Global L2 norm over every trainable gradient + clip coefficient min(1, max_norm / (norm + 1e-6)), kept on
        the device (fsdp.py:238-240 → FSDP.clip_grad_norm_). Each rank sums its own slices; the partial sums are
        all-reduced. Returns the device scalar total norm."""
        st, lay, nb = self.store, self.store.layout, self.store.blocks_per_bucket
        for i, b in enumerate(lay.buckets):
            T.sumsq_partial(st.reduced_grad(b), st.partial[i * nb:(i + 1) * nb])
        self.comm.all_reduce_sum(st.partial)
        T.clip_coef(st.partial, self.max_grad_norm, st.norm_coef)
        return st.norm_coef[0]

    def optimizer_step(self, lr: float, graph: bool = False) -> None:
        """AdamW on this rank's slice of the fp32 masters (decay classes per fsdp.py:200-212), all-gather of the updated
        bf16 slices, re-pack into the forward / dgrad layouts."""
        st, lay = self.store, self.store.layout
        st.step_count += 1
        main = torch.cuda.current_stream()
        for i in comm_order(lay.buckets):
            b = lay.buckets[i]
            lo, hi = lay.shard_range(b)
            sl = slice(lay.local_offset(b), lay.local_offset(b) + hi - lo)
            sharded = b.key in st.sharded_keys                     # parameter-sharded: the rank's bf16 slice IS the parameter
            T.adamw(st.master[sl], st.m[sl], st.v[sl], st.reduced_grad(b), st.step_count, lr, betas=self.betas, eps=self.eps,
                    weight_decay=self.weight_decay if b.decay else 0.0, norm_coef=st.norm_coef,
                    p_bf16=st.own_slice(b) if sharded else st.stage_view(lo, hi - lo))
            if self.comm.active and not sharded:                   # gather bucket i while AdamW runs on bucket i+1
                ev = torch.cuda.Event()
                ev.record(main)
                self._comm_stream.wait_event(ev)
                with torch.cuda.stream(self._comm_stream):
                    self.comm.all_gather_params(st.stage_view(b.offset, b.numel), b)
        if self.comm.active:
            main.wait_stream(self._comm_stream)
        self._replay("repack", self.repack_ops, graph)

    def step(self, lr: float, graph: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """forward → backward → clip → AdamW. Returns (loss, grad norm) as device scalars (no host sync here)."""
        loss = self.forward(graph)
        self.backward(graph)
        norm = self.clip_grad_norm()
        self.optimizer_step(lr, graph)
        return loss, norm
