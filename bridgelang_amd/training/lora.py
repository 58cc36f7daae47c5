"""LoRA adapters for the fine-tuning loop of vla-scripts/finetune.py:174-189 (PEFT `LoraConfig(r, lora_alpha=min(r, 16),
lora_dropout=0.0, target_modules="all-linear", init_lora_weights="gaussian")`): every nn.Linear except lm_head gets
y = W x + (alpha / r) · B (A x), A ~ N(0, 1/r), B = 0, only A and B train (AdamW, finetune.py:188).

Device layout: one adapter per packed GEMM group of weights.py. Groups that fuse several nn.Linear (q‖k‖v, interleaved
gate/up) hold their members' adapters side by side — A_all [m·rp, K] stacks the members' A, B_all [N, m·rp] is block
structured (member j's rows only see columns j·rp …), and the off-block entries are kept at zero by masking their
gradients (bl_lora_block_mask_f32). The rank is zero-padded to rp = 64 (the GEMM kernels' K granularity); padded rows /
columns have zero gradients by construction and stay zero.

Forward of one linear (training/step.py::_lin): t = x·Aᵀ is written into spare columns of x's buffer and the layer is ONE
GEMM over K + R, y = bf16([x | t]·[W | s·B]ᵀ) — the rank-R update is accumulated in fp32 with the base product and rounded
once. PEFT under bf16 autocast rounds lora_B's output, its product with the scaling and the sum with the base result
separately; the two differ by at most one bf16 ulp of y, but the single rounding does not absorb updates that are below
half an ulp of the base output (right after initialisation, B ≈ 0, PEFT's order loses most of them). s·B is stored as
bf16(s·B) (exact for the reference's s = 0.5).
Backward: dt = dy·(s·B) into spare columns of dy's buffer, dx = [dy | dt]·[Wᵀ | Aᵀ]ᵀ as one GEMM; dB = (s·tᵀ·dy)ᵀ and
dA = dtᵀ·x through the small-output TN GEMM (bl_gemm_tn_small_bf16: dy / x read once, untransposed).
`lora_dropout` = p > 0 (round 4): t = dropout(x)·Aᵀ from a masked copy of x (bl_dropout_bf16: counter-based mask, recomputed in
the backward pass), dA = dtᵀ·dropout(x), and the fused input gradient dy·W + dt·A is corrected to dy·W + mask/(1-p) ⊙ (dt·A)
with u = dt·A from one more rank-R GEMM (bl_dropout_grad_fix_bf16). p = 0 plans are unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from .. import ops
from ..weights import PackedGroup, VLAWeights

RP = 64          # padded rank


@dataclass(eq=False)
class Adapter:
    group: PackedGroup
    modules: List[str]            # HF module paths of the fused members (without ".weight")
    mode: str                     # "single" | "concat" | "interleave"
    r: int
    A: torch.Tensor               # bf16 [m*RP, k] row-major (live compute copy, AdamW's bf16 output)
    B: torch.Tensor               # bf16 [n, m*RP]
    A_p: torch.Tensor             # packed
    B_p: torch.Tensor
    shapes: List[Tuple[int, int]] = field(default_factory=list)     # per member (out_features, in_features)
    index: int = 0

    @property
    def R(self) -> int:
        return self.A.shape[0]

    def member_rows(self, j: int) -> torch.Tensor:
        """Row indices of member j inside the group's logical [n, k] matrix."""
        m, out = len(self.modules), self.shapes[j][0]
        if self.mode == "interleave":
            return torch.arange(out) * m + j
        base = (self.group.n // m) * j if self.mode == "concat" else 0
        return torch.arange(out) + base


class LoraAdapters:
    """All adapters of a model + their PEFT-named state dict."""

    def __init__(self, w: VLAWeights, r: int = 32, alpha: Optional[int] = None, seed: int = 0, dropout: float = 0.0):
        """`dropout` = PEFT's `lora_dropout` (finetune.py:101,177): in training the adapter branch sees nn.Dropout(p)(x). One mask
        per adapted GEMM and step: PEFT draws an independent mask for every module, here the modules fused into one GEMM
        (q / k / v; gate / up) share theirs — each module's own statistics are PEFT's, only the correlation between the
        fused modules' masks differs (training/step.py::_lin)."""
        if r > RP:
            raise ValueError(f"LoRA rank {r} exceeds the padded rank {RP}")
        if not (0.0 <= dropout < 1.0):
            raise ValueError(f"lora_dropout {dropout} outside [0, 1)")
        self.w, self.r, self.dropout = w, r, float(dropout)
        self.alpha = min(r, 16) if alpha is None else alpha
        self.scaling = self.alpha / r
        dev = w.embed.device
        specs = w._specs()
        self.adapters: List[Adapter] = []
        self.by_packed: Dict[int, Adapter] = {}
        z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
        for g in w.groups:
            first = g.members[0]
            if first == "language_model.lm_head.weight" or first.endswith("patch_embed.proj.weight"):
                continue                     # PEFT all-linear skips the output layer; the patch embedding is a Conv2d
            m = len(g.members)
            pl0 = w.placements[first]
            mode = "single" if m == 1 else ("interleave" if pl0.ld == m * pl0.cols else "concat")
            ad = Adapter(g, [n[:-len(".weight")] for n in g.members], mode, r, z(m * RP, g.k), z(g.n, m * RP),
                         z(m * RP // 16, g.k // 32, 64, 8), z(g.n // 16, m * RP // 32, 64, 8),
                         [tuple(specs[n].shape) for n in g.members])
            ad.index = len(self.adapters)
            self.adapters.append(ad)
            self.by_packed[g.packed.data_ptr()] = ad
        self.init_gaussian(seed)

    def get(self, packed: torch.Tensor) -> Optional[Adapter]:
        return self.by_packed.get(packed.data_ptr())

    # ---- values ----
    def init_gaussian(self, seed: int = 0) -> "LoraAdapters":
        """PEFT init_lora_weights="gaussian": A ~ N(0, 1/r), B = 0."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        sd = {}
        for ad in self.adapters:
            for j, mod in enumerate(ad.modules):
                out, inp = ad.shapes[j]
                sd[f"base_model.model.{mod}.lora_A.weight"] = torch.randn(self.r, inp, generator=gen) / self.r
                sd[f"base_model.model.{mod}.lora_B.weight"] = torch.zeros(out, self.r)
        return self.load_state_dict(sd)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> "LoraAdapters":
        for ad in self.adapters:
            A = torch.zeros(ad.A.shape, dtype=torch.float32)
            B = torch.zeros(ad.B.shape, dtype=torch.float32)
            for j, mod in enumerate(ad.modules):
                out, inp = ad.shapes[j]
                a, b = sd[f"base_model.model.{mod}.lora_A.weight"], sd[f"base_model.model.{mod}.lora_B.weight"]
                if tuple(a.shape) != (self.r, inp) or tuple(b.shape) != (out, self.r):
                    raise ValueError(f"{mod}: adapter shapes {tuple(a.shape)}, {tuple(b.shape)}")
                A[j * RP:j * RP + self.r, :inp] = a.float()
                B[ad.member_rows(j), j * RP:j * RP + self.r] = b.float()
            ad.A.copy_(A.to(torch.bfloat16))
            ad.B.copy_(B.to(torch.bfloat16))
        self.repack()
        return self

    def state_dict(self, masters: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """PEFT adapter-file names (`adapter_model`): base_model.model.<module>.lora_{A,B}.weight. `masters` (name →
        fp32 [m*RP, k] / [n, m*RP] from the ParamStore) replaces the live bf16 values when given."""
        out = {}
        for i, ad in enumerate(self.adapters):
            A = (masters[f"lora.{i}.A"] if masters else ad.A).float().cpu().view(ad.A.shape)
            B = (masters[f"lora.{i}.B"] if masters else ad.B).float().cpu().view(ad.B.shape)
            for j, mod in enumerate(ad.modules):
                o, inp = ad.shapes[j]
                out[f"base_model.model.{mod}.lora_A.weight"] = A[j * RP:j * RP + self.r, :inp].clone()
                out[f"base_model.model.{mod}.lora_B.weight"] = B[ad.member_rows(j), j * RP:j * RP + self.r].clone()
        return out

    def repack(self) -> None:
        for ad in self.adapters:
            ops.pack_weight(ad.A, ad.A_p)
            ops.pack_weight(ad.B, ad.B_p)

    def plain_units(self) -> List[Tuple[str, torch.Tensor, bool, str]]:
        """(name, live bf16 tensor, weight-decayed, bucket) for the ParamStore: finetune.py:188 builds one AdamW group
        (torch default weight_decay 0.01 on every adapter tensor)."""
        from .sharding import bucket_key
        out = []
        for i, ad in enumerate(self.adapters):
            bk = "lora." + bucket_key(ad.group.members[0])
            out.append((f"lora.{i}.A", ad.A.view(-1), True, bk))
            out.append((f"lora.{i}.B", ad.B.view(-1), True, bk))
        return out

    def n_params(self) -> int:
        return sum(self.r * (o + i) for ad in self.adapters for (o, i) in ad.shapes)

    def merged_state_dict(self) -> Dict[str, torch.Tensor]:
        """`merge_and_unload()` (finetune.py:331-341): HF-named base weights with W += scaling · B·A (export glue)."""
        sd = self.w.state_dict()
        ad_sd = self.state_dict()
        for ad in self.adapters:
            for mod in ad.modules:
                a = ad_sd[f"base_model.model.{mod}.lora_A.weight"].to(sd[mod + ".weight"].device)
                b = ad_sd[f"base_model.model.{mod}.lora_B.weight"].to(a.device)
                wt = sd[mod + ".weight"]
                sd[mod + ".weight"] = (wt.float() + self.scaling * (b @ a)).to(wt.dtype)
        return sd
