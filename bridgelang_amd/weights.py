"""Model dimensions, HF tensor naming, and the packed HBM layout of the OpenVLA weights.

On-disk naming follows the reference's HF layout (vla-scripts/extern/convert_openvla_weights_to_hf.py:73-115):
`vision_backbone.featurizer.*` (DINOv2), `vision_backbone.fused_featurizer.*` (SigLIP), `projector.fc{1,2,3}.*`,
`language_model.*`, LayerScale as `.scale_factor` (modeling_prismatic.py:52-59).

Device layout (bf16 arenas of 256-byte aligned sub-tensors, 15 GB at 7B — a fraction of the 288 GB HBM3E; four allocations: the
decoder layers' GEMM weights, the vision units', the head units' — projector, token embeddings, lm_head — each releasable under
parameter-sharded training, and the small plain tensors):
  * K dimensions padded to multiples of 64 (patch-embed 588→640, SigLIP MLP 4304→4352) so GEMM tiles never straddle a
    row end; the pad is zero and never leaves HBM/LDS.
  * Llama q/k/v stacked into one [3D, D] matrix; gate/up interleaved row-wise (2j = gate_j, 2j+1 = up_j) so the SwiGLU
    epilogue finds each (gate, up) pair in one lane.
  * only blocks up to the tap (n = depth-2, modeling_prismatic.py:85-87,99-101) are in the arena: the last ViT block,
    final norm and SigLIP attention-pool head never run on this path. They are kept as pass-through
    tensors (`passthrough_specs`) so that state dicts written here carry every key the reference loads strictly.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterator, List, Optional, Tuple

import torch


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


@dataclass(frozen=True)
class TowerDims:
    prefix: str            # HF state-dict prefix
    dim: int
    depth: int             # blocks in the checkpoint; blocks 0..depth-2 run
    heads: int
    mlp: int
    n_prefix: int          # cls + register tokens (DINOv2: 5, SigLIP: 0)
    layerscale: bool
    chan0: int             # first channel of the 6-channel pixel stack

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    @property
    def n_run(self) -> int:
        return self.depth - 1

    @property
    def mlp_pad(self) -> int:
        return _pad64(self.mlp)

    @property
    def tokens(self) -> int:
        return 256 + self.n_prefix


@dataclass(frozen=True)
class VLADims:
    dino: TowerDims
    siglip: TowerDims
    llm_dim: int
    llm_layers: int
    llm_heads: int
    llm_inter: int
    vocab: int = 32064            # 32000 + pad_to_multiple_of 64 (configuration_prismatic.py:85-86, llama2.py:74-76)
    rms_eps: float = 1e-6         # HF-path default LlamaConfig (native path uses 1e-5)
    rope_theta: float = 10000.0
    max_pos: int = 2048
    ln_eps: float = 1e-6
    n_patches: int = 256
    patch_k: int = 588            # 3*14*14
    name: str = "openvla-7b"

    @property
    def vision_dim(self) -> int:
        return self.dino.dim + self.siglip.dim

    @property
    def head_dim(self) -> int:
        return self.llm_dim // self.llm_heads


def openvla_7b_dims() -> VLADims:
    """prism-dinosiglip-224px + Llama-2-7B (configuration_prismatic.py:36)."""
    return VLADims(
        dino=TowerDims("vision_backbone.featurizer", 1024, 24, 16, 4096, 5, True, 0),
        siglip=TowerDims("vision_backbone.fused_featurizer", 1152, 27, 16, 4304, 0, False, 3),
        llm_dim=4096, llm_layers=32, llm_heads=32, llm_inter=11008)


def prism_13b_dims() -> VLADims:
    """BASELINE configs[4] composition: `dinosiglip-vit-so-224px` + `llama2-13b-pure` + fused-gelu-mlp (not a registered
    reference model id — SURVEY §8d cfg 5 — but every part is in the registries): hidden 5120, 40 layers / heads, 13824."""
    return VLADims(
        dino=TowerDims("vision_backbone.featurizer", 1024, 24, 16, 4096, 5, True, 0),
        siglip=TowerDims("vision_backbone.fused_featurizer", 1152, 27, 16, 4304, 0, False, 3),
        llm_dim=5120, llm_layers=40, llm_heads=40, llm_inter=13824, name="prism-dinosiglip-224px+13b")


def tiny_dims(llm_layers: int = 2, depth: int = 3) -> VLADims:
    """Reduced widths with the same structure (head_dim 64/72/128, a ragged SigLIP MLP width) for oracle-speed tests."""
    return VLADims(
        dino=TowerDims("vision_backbone.featurizer", 256, depth, 4, 1024, 5, True, 0),
        siglip=TowerDims("vision_backbone.fused_featurizer", 576, depth, 8, 1072, 0, False, 3),
        llm_dim=512, llm_layers=llm_layers, llm_heads=4, llm_inter=1536, name="openvla-tiny")


# ---- synthetic distributions -----------------------------------------------------------------------------------
IRWIN_HALL_SD = 37837.2265625   # sd of the sum of four uniform 16-bit integers: 65536/sqrt(3) (fp32-exact constant)


@dataclass(frozen=True)
class TensorSpec:
    name: str
    shape: Tuple[int, ...]
    mean: float
    std: float
    base: Optional[str] = None     # overlay specs only: the tensor whose rows [row0, row0 + shape[0]) this block replaces
    row0: int = 0


RECIPES = ("init", "decisive", "margin")
MARGIN_BRANCH_SCALE = 1.0 / 16.0     # "margin": residual branches this much smaller again than "decisive"'s …
MARGIN_LIVE_SCALE = 1.0              # … except the last decoder layer's attention output projection: this x the init std
MARGIN_LIVE_MLP_SCALE = 1.0          # … and its MLP down projection: this x the init std
MARGIN_BOOSTED_ROWS = 256            # all 256 action bins carry the 6x lm_head rows (32 were tried: larger relative gaps, but the
                                     # greedy map token -> next token on 32 symbols falls into a fixed point within a step or two)


def tensor_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ ((seed * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF


def _tower_specs(t: TowerDims) -> Iterator[TensorSpec]:
    p, D = t.prefix, t.dim
    if t.n_prefix:
        yield TensorSpec(f"{p}.cls_token", (1, 1, D), 0.0, 0.02)
        yield TensorSpec(f"{p}.reg_token", (1, t.n_prefix - 1, D), 0.0, 0.02)
    yield TensorSpec(f"{p}.pos_embed", (1, 256, D), 0.0, 0.02)
    yield TensorSpec(f"{p}.patch_embed.proj.weight", (D, 3, 14, 14), 0.0, 0.02)
    yield TensorSpec(f"{p}.patch_embed.proj.bias", (D,), 0.0, 0.02)
    for i in range(t.n_run):
        yield from _block_specs(t, i)


def _block_specs(t: TowerDims, i: int) -> Iterator[TensorSpec]:
    p, D = t.prefix, t.dim
    if True:
        b = f"{p}.blocks.{i}"
        yield TensorSpec(f"{b}.norm1.weight", (D,), 1.0, 0.02)
        yield TensorSpec(f"{b}.norm1.bias", (D,), 0.0, 0.02)
        yield TensorSpec(f"{b}.attn.qkv.weight", (3 * D, D), 0.0, 0.02)
        yield TensorSpec(f"{b}.attn.qkv.bias", (3 * D,), 0.0, 0.02)
        yield TensorSpec(f"{b}.attn.proj.weight", (D, D), 0.0, 0.02)
        yield TensorSpec(f"{b}.attn.proj.bias", (D,), 0.0, 0.02)
        if t.layerscale:
            yield TensorSpec(f"{b}.ls1.scale_factor", (D,), 0.1, 0.02)
        yield TensorSpec(f"{b}.norm2.weight", (D,), 1.0, 0.02)
        yield TensorSpec(f"{b}.norm2.bias", (D,), 0.0, 0.02)
        yield TensorSpec(f"{b}.mlp.fc1.weight", (t.mlp, D), 0.0, 0.02)
        yield TensorSpec(f"{b}.mlp.fc1.bias", (t.mlp,), 0.0, 0.02)
        yield TensorSpec(f"{b}.mlp.fc2.weight", (D, t.mlp), 0.0, 0.02)
        yield TensorSpec(f"{b}.mlp.fc2.bias", (D,), 0.0, 0.02)
        if t.layerscale:
            yield TensorSpec(f"{b}.ls2.scale_factor", (D,), 0.1, 0.02)


def passthrough_specs(d: VLADims) -> List[TensorSpec]:
    """Tensors of the checkpoint that the path never executes but that the reference's modules own and load STRICTLY
    (prismatic.py:113-116, convert_openvla_weights_to_hf.py:236): the last block of each timm tower (the tap is block
    depth-2, modeling_prismatic.py:85-87,99-101), the towers' final `norm`, and SigLIP's attention-pool head
    (`global_pool='map'`, timm 0.9.10 AttentionPoolLatent — † names from knowledge of timm, unverifiable offline, SURVEY
    App. A.1). They are held beside the arena, round-tripped by load_state_dict / state_dict, and never read by a kernel."""
    out: List[TensorSpec] = []
    for t in (d.dino, d.siglip):
        out += list(_block_specs(t, t.depth - 1))
        out += [TensorSpec(f"{t.prefix}.norm.weight", (t.dim,), 1.0, 0.02), TensorSpec(f"{t.prefix}.norm.bias", (t.dim,), 0.0, 0.02)]
    t, ap = d.siglip, f"{d.siglip.prefix}.attn_pool"
    D = t.dim
    out += [TensorSpec(f"{ap}.latent", (1, 1, D), 0.0, 0.02),
            TensorSpec(f"{ap}.q.weight", (D, D), 0.0, 0.02), TensorSpec(f"{ap}.q.bias", (D,), 0.0, 0.02),
            TensorSpec(f"{ap}.kv.weight", (2 * D, D), 0.0, 0.02), TensorSpec(f"{ap}.kv.bias", (2 * D,), 0.0, 0.02),
            TensorSpec(f"{ap}.proj.weight", (D, D), 0.0, 0.02), TensorSpec(f"{ap}.proj.bias", (D,), 0.0, 0.02),
            TensorSpec(f"{ap}.norm.weight", (D,), 1.0, 0.02), TensorSpec(f"{ap}.norm.bias", (D,), 0.0, 0.02),
            TensorSpec(f"{ap}.mlp.fc1.weight", (t.mlp, D), 0.0, 0.02), TensorSpec(f"{ap}.mlp.fc1.bias", (t.mlp,), 0.0, 0.02),
            TensorSpec(f"{ap}.mlp.fc2.weight", (D, t.mlp), 0.0, 0.02), TensorSpec(f"{ap}.mlp.fc2.bias", (D,), 0.0, 0.02)]
    return out


def tensor_specs(d: VLADims, recipe: str = "init") -> List[TensorSpec]:
    """Every tensor the path reads, under its HF state-dict name, with the synthetic distribution that stands in for the
    checkpoint (normal(0, 0.02) weights as in modeling_prismatic.py:185-205; norm scales around 1; LayerScale around 0.1
    so the DINOv2 branches stay visible in the output — SURVEY §8d's 1e-5 init would hide them from every parity test).

    recipe="init" is the bench checkpoint. recipe="decisive" is a second synthetic checkpoint for id-exact
    parity tests at full size: a freshly initialised 32-layer decoder is chaotic (every residual branch is as large as
    the stream it is added to, so bf16 rounding noise grows to percents of the logit scale and the 32064 logits are flat),
    which a trained checkpoint is not. It differs from "init" in three std values: residual-branch output projections
    (attn.proj / mlp.fc2 / o_proj / down_proj) are scaled by 1/sqrt(2·depth) (GPT-2 style), token embeddings have unit
    std (they then carry the stream), and — as an overlay, synthetic_overlays() — the lm_head rows of the 256 action
    tokens are 6x larger, so greedy decoding lands in the action vocabulary as a fine-tuned OpenVLA does.

    recipe="margin" is "decisive" with every residual branch a further 16x smaller (MARGIN_BRANCH_SCALE) EXCEPT the two
    branches of the LAST decoder layer, whose o_proj and down_proj keep the init scale (MARGIN_LIVE_SCALE /
    MARGIN_LIVE_MLP_SCALE). Between two correct fp32 summation orders the relative difference of the residual stream
    grows by about 0.0024 x (branch / stream ratio) x (relative difference of the branch input) per residual add — each
    re-rounding of the stream to bf16 turns the branch's small difference into rare whole-ulp flips (DESIGN.md §4 "noise
    floor") — so 62 contractive adds keep the stream's difference at a fraction of a percent, and two large branches at
    the very end, fed by that quiet stream, add the input- and step-dependent content (the last position attends over the
    256 image-patch rows and the prompt; the MLP hashes token and context) without much noise. On this checkpoint the
    logit noise is 1.4 % of the logit scale (half of "decisive"'s) while the greedy ids still differ from sequence to
    sequence and from step to step, and sequences whose oracle top-2 gap is >= 3x that noise at all 7 steps exist (about
    one candidate in 35) and can be selected (tests/golden/make_margin_b16.py): the fixture on which the WHOLE [16, 7] id
    matrix must be bit-exact. Layers 0..30 contribute little to the result here; their sensitivity is what the "init" /
    "decisive" fixtures and the per-op full-size tests cover."""
    if recipe not in RECIPES:
        raise ValueError(f"unknown synthetic recipe {recipe!r}")
    out = list(_tower_specs(d.dino)) + list(_tower_specs(d.siglip))
    V, P, L = d.vision_dim, 4 * d.vision_dim, d.llm_dim
    out += [TensorSpec("projector.fc1.weight", (P, V), 0.0, 0.02), TensorSpec("projector.fc1.bias", (P,), 0.0, 0.02),
            TensorSpec("projector.fc2.weight", (L, P), 0.0, 0.02), TensorSpec("projector.fc2.bias", (L,), 0.0, 0.02),
            TensorSpec("projector.fc3.weight", (L, L), 0.0, 0.02), TensorSpec("projector.fc3.bias", (L,), 0.0, 0.02)]
    lm = "language_model.model"
    out.append(TensorSpec(f"{lm}.embed_tokens.weight", (d.vocab, L), 0.0, 0.02))
    for i in range(d.llm_layers):
        b = f"{lm}.layers.{i}"
        out.append(TensorSpec(f"{b}.input_layernorm.weight", (L,), 1.0, 0.02))
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            out.append(TensorSpec(f"{b}.self_attn.{n}.weight", (L, L), 0.0, 0.02))
        out.append(TensorSpec(f"{b}.post_attention_layernorm.weight", (L,), 1.0, 0.02))
        out.append(TensorSpec(f"{b}.mlp.gate_proj.weight", (d.llm_inter, L), 0.0, 0.02))
        out.append(TensorSpec(f"{b}.mlp.up_proj.weight", (d.llm_inter, L), 0.0, 0.02))
        out.append(TensorSpec(f"{b}.mlp.down_proj.weight", (L, d.llm_inter), 0.0, 0.02))
    out.append(TensorSpec(f"{lm}.norm.weight", (L,), 1.0, 0.02))
    out.append(TensorSpec("language_model.lm_head.weight", (d.vocab, L), 0.0, 0.02))
    if recipe in ("decisive", "margin"):
        extra = MARGIN_BRANCH_SCALE if recipe == "margin" else 1.0
        depth = {d.dino.prefix: d.dino.n_run, d.siglip.prefix: d.siglip.n_run, "language_model": d.llm_layers}

        live = {f"language_model.model.layers.{d.llm_layers - 1}.self_attn.o_proj.weight": MARGIN_LIVE_SCALE,
                f"language_model.model.layers.{d.llm_layers - 1}.mlp.down_proj.weight": MARGIN_LIVE_MLP_SCALE}

        def adjust(sp: TensorSpec) -> TensorSpec:
            if recipe == "margin" and sp.name in live:
                return TensorSpec(sp.name, sp.shape, sp.mean, sp.std * live[sp.name])
            if sp.name.endswith((".attn.proj.weight", ".mlp.fc2.weight", ".o_proj.weight", ".down_proj.weight")):
                n = next(v for k, v in depth.items() if sp.name.startswith(k))
                return TensorSpec(sp.name, sp.shape, sp.mean, sp.std * extra / (2.0 * n) ** 0.5)
            if sp.name.endswith("embed_tokens.weight"):
                return TensorSpec(sp.name, sp.shape, 0.0, 1.0)
            return sp
        out = [adjust(sp) for sp in out]
    return out


N_ACTION_TOKENS = 256      # OpenVLAConfig.n_action_bins (configuration_prismatic.py:134): ids vocab_size-256 .. vocab_size-1
TOKENIZER_VOCAB = 32000    # Llama-2 tokenizer size; the padded embedding has d.vocab = 32064 rows


def synthetic_overlays(d: VLADims, recipe: str = "init") -> List[TensorSpec]:
    """Row blocks generated separately and written over part of a base tensor (synthetic checkpoints only; they are not
    tensors of the state dict). "decisive" / "margin": the lm_head rows of the action tokens."""
    if recipe not in ("decisive", "margin") or d.vocab < TOKENIZER_VOCAB:
        return []
    n = N_ACTION_TOKENS if recipe == "decisive" else MARGIN_BOOSTED_ROWS
    return [TensorSpec("language_model.lm_head.weight#action_rows", (n, d.llm_dim), 0.0, 0.12,
                       base="language_model.lm_head.weight", row0=TOKENIZER_VOCAB - n)]


# ---- packed device layout --------------------------------------------------------------------------------------
@dataclass
class Placement:
    """Where one HF tensor lives: a [rows, cols] block at element `offset` (leading dim `ld`) of either a plain
    row-major device tensor (`dst`) or — for GEMM weights — of the logical [N, Kpad] matrix of a packed group."""
    dst: Optional[torch.Tensor]
    offset: int
    rows: int
    cols: int
    ld: int
    group: Optional["PackedGroup"] = None


@dataclass
class PackedGroup:
    """One GEMM weight in fragment-major layout [N/16, Kpad/32, 64, 8] (include/bridgelang_hip.h, "weight layout")
    together with the HF tensors that make up its logical [N, Kpad] matrix (q‖k‖v, interleaved gate/up, K padding)."""
    packed: torch.Tensor
    n: int
    k: int
    members: List[str] = field(default_factory=list)


def _block_view(flat: torch.Tensor, pl: "Placement") -> torch.Tensor:
    """[rows, cols] strided view of a placement (as_strided offsets are absolute within the storage)."""
    return torch.as_strided(flat, (pl.rows, pl.cols), (pl.ld, 1), flat.storage_offset() + pl.offset)


def _pack(staging: torch.Tensor, out: torch.Tensor) -> None:
    if staging.is_cuda:
        from . import ops
        ops.pack_weight(staging, out)          # bl_pack_weight_bf16
    else:                                      # CPU: layout tests only (no kernels run on CPU)
        n, k = staging.shape
        out.view(n // 16, k // 32, 4, 16, 8).copy_(staging.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4))


def _unpack(packed: torch.Tensor) -> torch.Tensor:
    nt, ks = packed.shape[0], packed.shape[1]
    return packed.view(nt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(nt * 16, ks * 32)


class Arena:
    """One flat bf16 allocation carved into 256-byte aligned tensors (flat parameter storage)."""

    def __init__(self, device: torch.device):
        self.device = device
        self._shapes: List[Tuple[Tuple[int, ...], int]] = []
        self._total = 0
        self.buf: Optional[torch.Tensor] = None

    def reserve(self, shape: Tuple[int, ...]) -> int:
        n = 1
        for s in shape:
            n *= s
        idx = len(self._shapes)
        self._shapes.append((shape, self._total))
        self._total += (n + 127) // 128 * 128
        return idx

    def commit(self) -> None:
        self.buf = torch.zeros(self._total, dtype=torch.bfloat16, device=self.device)

    def view(self, idx: int) -> torch.Tensor:
        shape, off = self._shapes[idx]
        n = 1
        for s in shape:
            n *= s
        return self.buf[off:off + n].view(shape)

    @property
    def nbytes(self) -> int:
        return self._total * 2


@dataclass
class BlockW:
    norm1_w: torch.Tensor; norm1_b: torch.Tensor
    qkv_w: torch.Tensor; qkv_b: torch.Tensor          # *_w: packed
    proj_w: torch.Tensor; proj_b: torch.Tensor
    ls1: Optional[torch.Tensor]
    norm2_w: torch.Tensor; norm2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor
    fc2_w: torch.Tensor; fc2_b: torch.Tensor
    ls2: Optional[torch.Tensor]


@dataclass
class TowerW:
    dims: TowerDims
    patch_w: torch.Tensor      # packed [D, 640]
    patch_b: torch.Tensor
    pos: torch.Tensor          # [256, D]
    prefix: Optional[torch.Tensor]   # [n_prefix, D] = cls ‖ registers
    blocks: List[BlockW]


@dataclass
class LayerW:
    ln1: torch.Tensor
    qkv_w: torch.Tensor        # packed [3D, D]  q ‖ k ‖ v
    o_w: torch.Tensor
    ln2: torch.Tensor
    gu_w: torch.Tensor         # packed [2I, D]  rows interleaved gate/up
    down_w: torch.Tensor       # packed [D, I]


@dataclass
class VLAWeights:
    dims: VLADims
    arena: Arena
    dino: TowerW
    siglip: TowerW
    fc1_w: torch.Tensor; fc1_b: torch.Tensor
    fc2_w: torch.Tensor; fc2_b: torch.Tensor
    fc3_w: torch.Tensor; fc3_b: torch.Tensor
    embed: torch.Tensor        # row-major [vocab, D] (gathered, never a GEMM operand)
    layers: List[LayerW]
    norm: torch.Tensor
    lm_head: torch.Tensor      # packed
    placements: Dict[str, Placement] = field(default_factory=dict)
    groups: List[PackedGroup] = field(default_factory=list)
    passthrough: Dict[str, torch.Tensor] = field(default_factory=dict)   # passthrough_specs(): stored, never executed
    layer_arena: Optional[Arena] = None      # the decoder layers' GEMM weights: a second allocation that parameter-sharded
                                             # training can give back (release_layer_weights)
    _layer_views: List[tuple] = field(default_factory=list)             # (layer index, field, group index, arena view index)
    # the other FSDP units (prismatic.py:285-306: every ViT block, the projector; fsdp.py:160-168: the root's embeddings and
    # lm_head) in two more releasable allocations: "vision" (patch embeddings + blocks of both towers) and "head"
    # (projector, token embeddings, lm_head). Parameter-sharded training re-points their tensors at gather slots too.
    unit_arenas: Dict[str, Arena] = field(default_factory=dict)
    _unit_views: List[tuple] = field(default_factory=list)  # (pool, unit key, path to the attribute, group index | None, view index, HF name | None)

    # ---- parameter-sharded training (training/step.py, shard_params): decoder-layer GEMM weights leave the device -------
    @property
    def layers_resident(self) -> bool:
        return self.layer_arena is not None and self.layer_arena.buf is not None

    def repoint_layer_weights(self, slots: List[Dict[str, torch.Tensor]]) -> None:
        """Point every decoder layer's packed GEMM weights (and their PackedGroups) at the gather slots: layer l uses
        slots[l % len(slots)]."""
        self.__dict__.pop("_fp8_layers", None)
        for l, key, gi, _ in self._layer_views:
            t = slots[l % len(slots)][key]
            assert tuple(t.shape) == tuple(getattr(self.layers[l], key).shape), (l, key)
            setattr(self.layers[l], key, t)
            self.groups[gi].packed = t

    def release_layer_weights(self, slots: List[Dict[str, torch.Tensor]]) -> int:
        """repoint_layer_weights + free the layers' allocation. Returns the bytes given back. Anything that still holds
        the old views (engine plans built before the call) keeps the storage alive: callers drop those first."""
        if not self.layers_resident:
            raise RuntimeError("decoder-layer weights are not resident")
        self.repoint_layer_weights(slots)
        self.layer_arena.buf = None
        return self.layer_arena.nbytes

    def restore_layer_weights(self) -> None:
        """Re-allocate the decoder layers' GEMM weights (zero-filled; the caller gathers and packs them)."""
        if self.layers_resident:
            return
        self.layer_arena.commit()
        for l, key, gi, idx in self._layer_views:
            t = self.layer_arena.view(idx)
            setattr(self.layers[l], key, t)
            self.groups[gi].packed = t

    # ---- the same for the vision / head units -----------------------------------------------------------------------------
    def _attr_of(self, path: tuple):
        obj = self
        for q in path[:-1]:
            obj = getattr(obj, q) if isinstance(q, str) else obj[q]
        return obj, path[-1]

    def pool_resident(self, pool: str) -> bool:
        a = self.unit_arenas.get(pool)
        return a is not None and a.buf is not None

    def unit_fields(self, pool: str) -> List[tuple]:
        """(unit key, attribute path, group index or None, HF name or None) of every tensor of the pool, allocation order."""
        return [(key, path, gi, name) for pl, key, path, gi, _, name in self._unit_views if pl == pool]

    def repoint_unit_weights(self, pool: str, views: Dict[tuple, torch.Tensor]) -> None:
        """views[(unit key, attribute path)] = the tensor (a gather-slot view of the same shape) that replaces it."""
        for pl, key, path, gi, _, name in self._unit_views:
            if pl != pool or (key, path) not in views:
                continue
            t = views[(key, path)]
            obj, attr = self._attr_of(path)
            assert tuple(t.shape) == tuple(getattr(obj, attr).shape), (key, path)
            setattr(obj, attr, t)
            if gi is not None:
                self.groups[gi].packed = t
            if name is not None:
                self.placements[name].dst = t

    def release_unit_weights(self, pool: str, views: Dict[tuple, torch.Tensor]) -> int:
        if not self.pool_resident(pool):
            raise RuntimeError(f"the {pool} weights are not resident")
        self.repoint_unit_weights(pool, views)
        self.unit_arenas[pool].buf = None
        return self.unit_arenas[pool].nbytes

    def restore_unit_weights(self, pool: str) -> None:
        """Re-allocate the pool (zero-filled; the caller gathers and packs)."""
        if self.pool_resident(pool):
            return
        arena = self.unit_arenas[pool]
        arena.commit()
        self.repoint_unit_weights(pool, {(key, path): arena.view(idx) for pl, key, path, gi, idx, name in self._unit_views if pl == pool})

    def _specs(self, recipe: str = "init") -> Dict[str, TensorSpec]:
        return {s.name: s for s in tensor_specs(self.dims, recipe)}

    # ---- filling ----
    def fill_synthetic(self, seed: int = 0, recipe: str = "init") -> "VLAWeights":
        """Fill every tensor on the device with the deterministic generator (bl_fill_synth_bf16_2d), then pack the GEMM
        weights (bl_pack_weight_bf16); the CPU oracle builds the identical tensors from the same (name, seed, mean,
        std) with oracle/synth.py. `recipe`: see tensor_specs()."""
        from . import ops
        self.__dict__.pop("_fp8_layers", None)        # derived e4m3 copies (engine.py) follow the bf16 weights
        specs = self._specs(recipe)
        overlays: Dict[str, List[TensorSpec]] = {}
        for ov in synthetic_overlays(self.dims, recipe):
            overlays.setdefault(ov.base, []).append(ov)

        def fill(flat, name):
            spec, pl = specs[name], self.placements[name]
            ops.fill_synth(flat[pl.offset:], tensor_seed(name, seed), spec.mean, spec.std / IRWIN_HALL_SD,
                           rows=pl.rows, cols=pl.cols, ld=pl.ld)
            for ov in overlays.get(name, ()):           # a row block of this tensor, generated on its own
                ops.fill_synth(flat[pl.offset + ov.row0 * pl.ld:], tensor_seed(ov.name, seed), ov.mean,
                               ov.std / IRWIN_HALL_SD, rows=ov.shape[0], cols=pl.cols, ld=pl.ld)
        for name, pl in self.placements.items():
            if pl.group is None:
                fill(pl.dst.view(-1), name)
        for g in self.groups:
            staging = torch.zeros(g.n, g.k, dtype=torch.bfloat16, device=g.packed.device)
            for name in g.members:
                fill(staging.view(-1), name)
            _pack(staging, g.packed)
        for sp in passthrough_specs(self.dims):
            ops.fill_synth(self.passthrough[sp.name].view(-1), tensor_seed(sp.name, seed), sp.mean, sp.std / IRWIN_HALL_SD)
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> "VLAWeights":
        """Pack an HF-named state dict (bf16/fp32 tensors on any device) into the device layout."""
        self.__dict__.pop("_fp8_layers", None)        # derived e4m3 copies (engine.py) follow the bf16 weights
        specs = self._specs()
        missing = [n for n in specs if n not in sd]
        if strict and missing:
            raise KeyError(f"state dict is missing {len(missing)} tensors, e.g. {missing[:3]}")

        def put(flat, name):
            spec, pl = specs[name], self.placements[name]
            src = sd[name]
            if tuple(src.shape) != spec.shape:
                raise ValueError(f"{name}: expected {spec.shape}, got {tuple(src.shape)}")
            _block_view(flat, pl).copy_(src.reshape(pl.rows, pl.cols).to(device=flat.device, dtype=torch.bfloat16))
        for name, pl in self.placements.items():
            if pl.group is None and name in sd:
                put(pl.dst.view(-1), name)
        for name, t in self.passthrough.items():      # never-executed tensors: kept for export (optional on input)
            if name in sd:
                if tuple(sd[name].shape) != tuple(t.shape):
                    raise ValueError(f"{name}: expected {tuple(t.shape)}, got {tuple(sd[name].shape)}")
                t.copy_(sd[name].to(device=t.device, dtype=torch.bfloat16))
        for g in self.groups:
            if not all(n in sd for n in g.members):
                continue
            staging = torch.zeros(g.n, g.k, dtype=torch.bfloat16, device=g.packed.device)
            for name in g.members:
                put(staging.view(-1), name)
            _pack(staging, g.packed)
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Unpack to HF names/shapes (device tensors, copies)."""
        specs = self._specs()
        out = {}
        for name, pl in self.placements.items():
            if pl.group is None:
                out[name] = _block_view(pl.dst.view(-1), pl).clone().reshape(specs[name].shape)
        for g in self.groups:
            staging = _unpack(g.packed).contiguous()
            for name in g.members:
                out[name] = _block_view(staging.view(-1), self.placements[name]).clone().reshape(specs[name].shape)
        out = {n: out[n] for n in specs}
        out.update({n: t.clone() for n, t in self.passthrough.items()})
        return out


def allocate(dims: VLADims, device: torch.device | str = "cuda") -> VLAWeights:
    """Reserve the arena and build the name → placement table. Tensors are zero until filled."""
    device = torch.device(device)
    arena = Arena(device)
    layer_arena = Arena(device)                  # decoder-layer GEMM weights (see VLAWeights.release_layer_weights)
    unit_arenas = {"vision": Arena(device), "head": Arena(device)}   # the other FSDP units (see VLAWeights.release_unit_weights)
    unit_views: List[tuple] = []
    pending_units: List[Tuple[str, int, Callable[[torch.Tensor], None]]] = []
    layer_views: List[tuple] = []
    pending: List[Tuple[int, Callable[[torch.Tensor], None]]] = []
    pending_layers: List[Tuple[int, Callable[[torch.Tensor], None]]] = []
    plain: Dict[str, tuple] = {}                 # name → (holder, key, offset, rows, cols, ld)
    grouped: Dict[str, tuple] = {}               # name → (group id, offset, rows, cols, ld)
    group_defs: List[tuple] = []                 # (holder, key, n, k)

    def dense(holder, key, shape):
        pending.append((arena.reserve(tuple(shape)), lambda x, h=holder, k=key: h.__setitem__(k, x)))

    def gemm_w(holder, key, n, k, layer: Optional[int] = None, unit: Optional[tuple] = None) -> int:
        """`unit` = (pool, bucket key, attribute path on the finished VLAWeights) for the vision / head units."""
        assert n % 16 == 0 and k % 64 == 0, (key, n, k)
        setter = lambda x, h=holder, kk=key: h.__setitem__(kk, x)
        if unit is not None:
            idx = unit_arenas[unit[0]].reserve((n // 16, k // 32, 64, 8))
            pending_units.append((unit[0], idx, setter))
            unit_views.append((unit[0], unit[1], unit[2], len(group_defs), idx, None))
        elif layer is None:
            pending.append((arena.reserve((n // 16, k // 32, 64, 8)), setter))
        else:
            idx = layer_arena.reserve((n // 16, k // 32, 64, 8))
            pending_layers.append((idx, setter))
            layer_views.append((layer, key, len(group_defs), idx))
        group_defs.append((holder, key, n, k))
        return len(group_defs) - 1

    def tower(t: TowerDims, attr: str) -> dict:
        h = {"blocks": [dict() for _ in range(t.n_run)]}
        p, D, Hp, kp = t.prefix, t.dim, t.mlp_pad, _pad64(dims.patch_k)
        tk = p.split(".")[1]                       # bucket keys as training/sharding.py::bucket_key names them
        g = gemm_w(h, "patch_w", D, kp, unit=("vision", f"vision.{tk}.stem", (attr, "patch_w")))
        grouped[f"{p}.patch_embed.proj.weight"] = (g, 0, D, dims.patch_k, kp)
        dense(h, "patch_b", (D,)); plain[f"{p}.patch_embed.proj.bias"] = (h, "patch_b", 0, 1, D, D)
        dense(h, "pos", (256, D)); plain[f"{p}.pos_embed"] = (h, "pos", 0, 256, D, D)
        if t.n_prefix:
            dense(h, "prefix", (t.n_prefix, D))
            plain[f"{p}.cls_token"] = (h, "prefix", 0, 1, D, D)
            plain[f"{p}.reg_token"] = (h, "prefix", D, t.n_prefix - 1, D, D)
        else:
            h["prefix"] = None
        for i in range(t.n_run):
            b, bn = h["blocks"][i], f"{p}.blocks.{i}"
            for key, hf in (("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("norm2_w", "norm2.weight"),
                            ("norm2_b", "norm2.bias"), ("proj_b", "attn.proj.bias"), ("fc2_b", "mlp.fc2.bias")):
                dense(b, key, (D,)); plain[f"{bn}.{hf}"] = (b, key, 0, 1, D, D)
            dense(b, "qkv_b", (3 * D,)); plain[f"{bn}.attn.qkv.bias"] = (b, "qkv_b", 0, 1, 3 * D, 3 * D)
            dense(b, "fc1_b", (Hp,)); plain[f"{bn}.mlp.fc1.bias"] = (b, "fc1_b", 0, 1, t.mlp, t.mlp)
            un = lambda f: ("vision", f"vision.{tk}.block{i:02d}", (attr, "blocks", i, f))
            g = gemm_w(b, "qkv_w", 3 * D, D, unit=un("qkv_w")); grouped[f"{bn}.attn.qkv.weight"] = (g, 0, 3 * D, D, D)
            g = gemm_w(b, "proj_w", D, D, unit=un("proj_w")); grouped[f"{bn}.attn.proj.weight"] = (g, 0, D, D, D)
            g = gemm_w(b, "fc1_w", Hp, D, unit=un("fc1_w")); grouped[f"{bn}.mlp.fc1.weight"] = (g, 0, t.mlp, D, D)
            g = gemm_w(b, "fc2_w", D, Hp, unit=un("fc2_w")); grouped[f"{bn}.mlp.fc2.weight"] = (g, 0, D, t.mlp, Hp)
            if t.layerscale:
                dense(b, "ls1", (D,)); plain[f"{bn}.ls1.scale_factor"] = (b, "ls1", 0, 1, D, D)
                dense(b, "ls2", (D,)); plain[f"{bn}.ls2.scale_factor"] = (b, "ls2", 0, 1, D, D)
            else:
                b["ls1"] = b["ls2"] = None
        return h

    hd, hs = tower(dims.dino, "dino"), tower(dims.siglip, "siglip")
    top: dict = {}
    V, P, L, I = dims.vision_dim, 4 * dims.vision_dim, dims.llm_dim, dims.llm_inter
    for key, hf, n, k in (("fc1_w", "projector.fc1.weight", P, V), ("fc2_w", "projector.fc2.weight", L, P),
                          ("fc3_w", "projector.fc3.weight", L, L)):
        g = gemm_w(top, key, n, k, unit=("head", "projector", (key,))); grouped[hf] = (g, 0, n, k, k)
    for key, hf, n in (("fc1_b", "projector.fc1.bias", P), ("fc2_b", "projector.fc2.bias", L),
                       ("fc3_b", "projector.fc3.bias", L)):
        dense(top, key, (n,)); plain[hf] = (top, key, 0, 1, n, n)
    lm = "language_model.model"
    # token embeddings: a plain [vocab, L] tensor (gathered by row), but one of the root FSDP unit's two big parameters
    idx = unit_arenas["head"].reserve((dims.vocab, L))
    pending_units.append(("head", idx, lambda x: top.__setitem__("embed", x)))
    unit_views.append(("head", "llm.embed", ("embed",), None, idx, f"{lm}.embed_tokens.weight"))
    plain[f"{lm}.embed_tokens.weight"] = (top, "embed", 0, dims.vocab, L, L)
    layer_h = [dict() for _ in range(dims.llm_layers)]
    for i, lh in enumerate(layer_h):
        bn = f"{lm}.layers.{i}"
        dense(lh, "ln1", (L,)); plain[f"{bn}.input_layernorm.weight"] = (lh, "ln1", 0, 1, L, L)
        dense(lh, "ln2", (L,)); plain[f"{bn}.post_attention_layernorm.weight"] = (lh, "ln2", 0, 1, L, L)
        g = gemm_w(lh, "qkv_w", 3 * L, L, layer=i)
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            grouped[f"{bn}.self_attn.{n}.weight"] = (g, j * L * L, L, L, L)
        g = gemm_w(lh, "o_w", L, L, layer=i); grouped[f"{bn}.self_attn.o_proj.weight"] = (g, 0, L, L, L)
        g = gemm_w(lh, "gu_w", 2 * I, L, layer=i)
        grouped[f"{bn}.mlp.gate_proj.weight"] = (g, 0, I, L, 2 * L)      # row 2j   = gate_j
        grouped[f"{bn}.mlp.up_proj.weight"] = (g, L, I, L, 2 * L)        # row 2j+1 = up_j
        g = gemm_w(lh, "down_w", L, I, layer=i); grouped[f"{bn}.mlp.down_proj.weight"] = (g, 0, L, I, I)
    dense(top, "norm", (L,)); plain[f"{lm}.norm.weight"] = (top, "norm", 0, 1, L, L)
    g = gemm_w(top, "lm_head", dims.vocab, L, unit=("head", "llm.lm_head", ("lm_head",)))
    grouped["language_model.lm_head.weight"] = (g, 0, dims.vocab, L, L)

    arena.commit()
    layer_arena.commit()
    for a in unit_arenas.values():
        a.commit()
    for pool, idx, setter in pending_units:
        setter(unit_arenas[pool].view(idx))
    for idx, setter in pending:
        setter(arena.view(idx))
    for idx, setter in pending_layers:
        setter(layer_arena.view(idx))

    def mk_tower(t: TowerDims, h: dict) -> TowerW:
        return TowerW(t, h["patch_w"], h["patch_b"], h["pos"], h["prefix"], [BlockW(**b) for b in h["blocks"]])

    w = VLAWeights(dims, arena, mk_tower(dims.dino, hd), mk_tower(dims.siglip, hs),
                   top["fc1_w"], top["fc1_b"], top["fc2_w"], top["fc2_b"], top["fc3_w"], top["fc3_b"], top["embed"],
                   [LayerW(**lh) for lh in layer_h], top["norm"], top["lm_head"])
    w.layer_arena, w._layer_views = layer_arena, layer_views
    w.unit_arenas, w._unit_views = unit_arenas, unit_views
    w.groups = [PackedGroup(holder[key], n, k) for holder, key, n, k in group_defs]
    for name, (holder, key, off, rows, cols, ld) in plain.items():
        w.placements[name] = Placement(holder[key], off, rows, cols, ld)
    for name, (gi, off, rows, cols, ld) in grouped.items():
        w.placements[name] = Placement(None, off, rows, cols, ld, w.groups[gi])
        w.groups[gi].members.append(name)
    # never-executed checkpoint tensors (≈ 45 M parameters, 90 MB): plain tensors beside the arena; norm scales start at 1
    for sp in passthrough_specs(dims):
        w.passthrough[sp.name] = torch.full(sp.shape, sp.mean, dtype=torch.bfloat16, device=device)
    return w
