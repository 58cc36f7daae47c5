"""Model dimensions, HF tensor naming, and the packed HBM layout of the OpenVLA weights.

On-disk naming follows the reference's HF layout (vla-scripts/extern/convert_openvla_weights_to_hf.py:73-115):
`vision_backbone.featurizer.*` (DINOv2), `vision_backbone.fused_featurizer.*` (SigLIP), `projector.fc{1,2,3}.*`,
`language_model.*`, LayerScale as `.scale_factor` (modeling_prismatic.py:52-59).

Device layout (one bf16 arena, 256-byte aligned sub-tensors; 15 GB at 7B — a fraction of the 288 GB HBM3E):
  * K dimensions padded to multiples of 64 (patch-embed 588→640, SigLIP MLP 4304→4352) so GEMM tiles never straddle a
    row end; the pad is zero and never leaves HBM/LDS.
  * Llama q/k/v stacked into one [3D, D] matrix; gate/up interleaved row-wise (2j = gate_j, 2j+1 = up_j) so the SwiGLU
    epilogue finds each (gate, up) pair in one lane.
  * only blocks up to the tap (n = depth-2, modeling_prismatic.py:85-87,99-101) are materialised: the last ViT block,
    final norm and SigLIP attention-pool head never run on this path.
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


def tensor_specs(d: VLADims) -> List[TensorSpec]:
    """Every tensor the path reads, under its HF state-dict name, with the synthetic distribution that stands in for the
    checkpoint (normal(0, 0.02) weights as in modeling_prismatic.py:185-205; norm scales around 1; LayerScale around 0.1
    so the DINOv2 branches stay visible in the output — SURVEY §8d's 1e-5 init would hide them from every parity test)."""
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
    return out


# ---- packed device layout --------------------------------------------------------------------------------------
@dataclass
class Placement:
    """Where one HF tensor lives inside a packed device tensor: a [rows, cols] block at `offset` with leading dim `ld`."""
    dst: torch.Tensor
    offset: int
    rows: int
    cols: int
    ld: int


def _block_view(pl: "Placement") -> torch.Tensor:
    """[rows, cols] strided view of a placement (as_strided offsets are absolute within the storage)."""
    flat = pl.dst.view(-1)
    return torch.as_strided(flat, (pl.rows, pl.cols), (pl.ld, 1), flat.storage_offset() + pl.offset)


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
    qkv_w: torch.Tensor; qkv_b: torch.Tensor
    proj_w: torch.Tensor; proj_b: torch.Tensor
    ls1: Optional[torch.Tensor]
    norm2_w: torch.Tensor; norm2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor
    fc2_w: torch.Tensor; fc2_b: torch.Tensor
    ls2: Optional[torch.Tensor]


@dataclass
class TowerW:
    dims: TowerDims
    patch_w: torch.Tensor      # [D, 640]
    patch_b: torch.Tensor
    pos: torch.Tensor          # [256, D]
    prefix: Optional[torch.Tensor]   # [n_prefix, D] = cls ‖ registers
    blocks: List[BlockW]


@dataclass
class LayerW:
    ln1: torch.Tensor
    qkv_w: torch.Tensor        # [3D, D]  q ‖ k ‖ v
    o_w: torch.Tensor
    ln2: torch.Tensor
    gu_w: torch.Tensor         # [2I, D]  rows interleaved gate/up
    down_w: torch.Tensor       # [D, I]


@dataclass
class VLAWeights:
    dims: VLADims
    arena: Arena
    dino: TowerW
    siglip: TowerW
    fc1_w: torch.Tensor; fc1_b: torch.Tensor
    fc2_w: torch.Tensor; fc2_b: torch.Tensor
    fc3_w: torch.Tensor; fc3_b: torch.Tensor
    embed: torch.Tensor
    layers: List[LayerW]
    norm: torch.Tensor
    lm_head: torch.Tensor
    placements: Dict[str, Placement] = field(default_factory=dict)

    # ---- filling ----
    def fill_synthetic(self, seed: int = 0) -> "VLAWeights":
        """Fill every tensor on the device with the deterministic generator (bl_fill_synth_bf16_2d); the CPU oracle
        builds the identical tensors from the same (name, seed, mean, std) with oracle/synth.py."""
        from . import ops
        for spec in tensor_specs(self.dims):
            pl = self.placements[spec.name]
            flat = pl.dst.view(-1)[pl.offset:]
            ops.fill_synth(flat, tensor_seed(spec.name, seed), spec.mean, spec.std / IRWIN_HALL_SD,
                           rows=pl.rows, cols=pl.cols, ld=pl.ld)
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> "VLAWeights":
        """Pack an HF-named state dict (bf16/fp32 tensors on any device) into the device layout."""
        missing = []
        for spec in tensor_specs(self.dims):
            if spec.name not in sd:
                missing.append(spec.name)
                continue
            src = sd[spec.name]
            if tuple(src.shape) != spec.shape:
                raise ValueError(f"{spec.name}: expected {spec.shape}, got {tuple(src.shape)}")
            pl = self.placements[spec.name]
            dst = _block_view(pl)
            dst.copy_(src.reshape(pl.rows, pl.cols).to(device=dst.device, dtype=torch.bfloat16))
        if strict and missing:
            raise KeyError(f"state dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Unpack to HF names/shapes (device tensors, copies)."""
        out = {}
        for spec in tensor_specs(self.dims):
            pl = self.placements[spec.name]
            src = _block_view(pl)
            out[spec.name] = src.clone().reshape(spec.shape)
        return out


def allocate(dims: VLADims, device: torch.device | str = "cuda") -> VLAWeights:
    """Reserve the arena and build the name → placement table. Tensors are zero until filled."""
    device = torch.device(device)
    arena = Arena(device)
    pending: List[Tuple[int, Callable[[torch.Tensor], None]]] = []
    placements: Dict[str, Placement] = {}

    def T(shape, setter):
        pending.append((arena.reserve(tuple(shape)), setter))

    def place(name, holder, key, offset, rows, cols, ld):
        placements[name] = (holder, key, offset, rows, cols, ld)   # resolved after commit

    holders: Dict[str, dict] = {}

    def tower(t: TowerDims) -> dict:
        h = {"blocks": [dict() for _ in range(t.n_run)]}
        p, D, Hp = t.prefix, t.dim, t.mlp_pad
        kp = _pad64(dims.patch_k)
        T((D, kp), lambda x: h.__setitem__("patch_w", x)); place(f"{p}.patch_embed.proj.weight", h, "patch_w", 0, D, dims.patch_k, kp)
        T((D,), lambda x: h.__setitem__("patch_b", x)); place(f"{p}.patch_embed.proj.bias", h, "patch_b", 0, 1, D, D)
        T((256, D), lambda x: h.__setitem__("pos", x)); place(f"{p}.pos_embed", h, "pos", 0, 256, D, D)
        if t.n_prefix:
            T((t.n_prefix, D), lambda x: h.__setitem__("prefix", x))
            place(f"{p}.cls_token", h, "prefix", 0, 1, D, D)
            place(f"{p}.reg_token", h, "prefix", D, t.n_prefix - 1, D, D)
        else:
            h["prefix"] = None
        for i in range(t.n_run):
            b, bn = h["blocks"][i], f"{p}.blocks.{i}"

            def reg(key, shape, hf, rows, cols, ld, b=b):
                T(shape, lambda x, b=b, key=key: b.__setitem__(key, x))
                place(hf, b, key, 0, rows, cols, ld)
            reg("norm1_w", (D,), f"{bn}.norm1.weight", 1, D, D); reg("norm1_b", (D,), f"{bn}.norm1.bias", 1, D, D)
            reg("qkv_w", (3 * D, D), f"{bn}.attn.qkv.weight", 3 * D, D, D); reg("qkv_b", (3 * D,), f"{bn}.attn.qkv.bias", 1, 3 * D, 3 * D)
            reg("proj_w", (D, D), f"{bn}.attn.proj.weight", D, D, D); reg("proj_b", (D,), f"{bn}.attn.proj.bias", 1, D, D)
            reg("norm2_w", (D,), f"{bn}.norm2.weight", 1, D, D); reg("norm2_b", (D,), f"{bn}.norm2.bias", 1, D, D)
            reg("fc1_w", (Hp, D), f"{bn}.mlp.fc1.weight", t.mlp, D, D); reg("fc1_b", (Hp,), f"{bn}.mlp.fc1.bias", 1, t.mlp, t.mlp)
            reg("fc2_w", (D, Hp), f"{bn}.mlp.fc2.weight", D, t.mlp, Hp); reg("fc2_b", (D,), f"{bn}.mlp.fc2.bias", 1, D, D)
            if t.layerscale:
                reg("ls1", (D,), f"{bn}.ls1.scale_factor", 1, D, D); reg("ls2", (D,), f"{bn}.ls2.scale_factor", 1, D, D)
            else:
                b["ls1"] = b["ls2"] = None
        return h

    holders["dino"], holders["siglip"] = tower(dims.dino), tower(dims.siglip)
    top: dict = {}
    V, P, L, I = dims.vision_dim, 4 * dims.vision_dim, dims.llm_dim, dims.llm_inter

    def reg_top(key, shape, hf, rows, cols, ld):
        T(shape, lambda x, key=key: top.__setitem__(key, x))
        place(hf, top, key, 0, rows, cols, ld)
    reg_top("fc1_w", (P, V), "projector.fc1.weight", P, V, V); reg_top("fc1_b", (P,), "projector.fc1.bias", 1, P, P)
    reg_top("fc2_w", (L, P), "projector.fc2.weight", L, P, P); reg_top("fc2_b", (L,), "projector.fc2.bias", 1, L, L)
    reg_top("fc3_w", (L, L), "projector.fc3.weight", L, L, L); reg_top("fc3_b", (L,), "projector.fc3.bias", 1, L, L)
    lm = "language_model.model"
    reg_top("embed", (dims.vocab, L), f"{lm}.embed_tokens.weight", dims.vocab, L, L)
    layer_h = [dict() for _ in range(dims.llm_layers)]
    for i, lh in enumerate(layer_h):
        bn = f"{lm}.layers.{i}"

        def regl(key, shape, lh=lh):
            T(shape, lambda x, lh=lh, key=key: lh.__setitem__(key, x))
        regl("ln1", (L,)); place(f"{bn}.input_layernorm.weight", lh, "ln1", 0, 1, L, L)
        regl("qkv_w", (3 * L, L))
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            place(f"{bn}.self_attn.{n}.weight", lh, "qkv_w", j * L * L, L, L, L)
        regl("o_w", (L, L)); place(f"{bn}.self_attn.o_proj.weight", lh, "o_w", 0, L, L, L)
        regl("ln2", (L,)); place(f"{bn}.post_attention_layernorm.weight", lh, "ln2", 0, 1, L, L)
        regl("gu_w", (2 * I, L))
        place(f"{bn}.mlp.gate_proj.weight", lh, "gu_w", 0, I, L, 2 * L)
        place(f"{bn}.mlp.up_proj.weight", lh, "gu_w", L, I, L, 2 * L)
        regl("down_w", (L, I)); place(f"{bn}.mlp.down_proj.weight", lh, "down_w", 0, L, I, I)
    reg_top("norm", (L,), f"{lm}.norm.weight", 1, L, L)
    reg_top("lm_head", (dims.vocab, L), "language_model.lm_head.weight", dims.vocab, L, L)

    arena.commit()
    for idx, setter in pending:
        setter(arena.view(idx))

    def mk_tower(t: TowerDims, h: dict) -> TowerW:
        return TowerW(t, h["patch_w"], h["patch_b"], h["pos"], h["prefix"], [BlockW(**b) for b in h["blocks"]])

    w = VLAWeights(dims, arena, mk_tower(dims.dino, holders["dino"]), mk_tower(dims.siglip, holders["siglip"]),
                   top["fc1_w"], top["fc1_b"], top["fc2_w"], top["fc2_b"], top["fc3_w"], top["fc3_b"], top["embed"],
                   [LayerW(**lh) for lh in layer_h], top["norm"], top["lm_head"])
    w.placements = {name: Placement(holder[key], off, rows, cols, ld)
                    for name, (holder, key, off, rows, cols, ld) in placements.items()}
    return w
