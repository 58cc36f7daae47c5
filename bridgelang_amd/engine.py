"""OpenVLA inference engine: static-shape op plans over the HIP kernels, replayed eagerly or as one HIP graph.

The reference's `predict_action` = HF `generate(max_new_tokens=7)` over `PrismaticForConditionalGeneration.forward`
(modeling_prismatic.py:506-536, 291-447): one multimodal prefill (vision towers → projector → splice → 32 Llama layers)
and 6 cached decode steps, with a host round trip per token. Here the whole sequence — prefill, 6 decode steps, the
greedy argmax of every step — is a fixed list of kernel launches on one stream with all intermediate buffers
preallocated (activations for B=16 are < 1 GB of the 288 GB HBM), so it can be captured once and replayed with no host
synchronisation until the 7 token ids are read back. Batched generation is an extension over the reference (batch 1
only, modeling_prismatic.py:326,460-463); per-sample results equal independent batch-1 calls.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .ops import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_F32_BF16R, EPI_NONE, EPI_RES, EPI_SWIGLU, Op
from .weights import TowerW, VLAWeights


def rope_tables(head_dim: int, max_pos: int, theta: float, device) -> tuple:
    """HF LlamaRotaryEmbedding on the host, exactly as the reference computes it (fp32 trig, cast to bf16), uploaded
    once: [max_pos, head_dim/2] cos and sin tables."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().to(torch.bfloat16).to(device), fr.sin().to(torch.bfloat16).to(device)


def _fp8_layers(weights: VLAWeights) -> list:
    """e4m3 copies (+ per-channel scales) of the Llama projection weights, quantised once per weight set from the packed
    bf16 arena (ops.quantize_weight_fp8) and kept on the weights object: shared by every engine built over these weights.
    (Quantised from the weights as they are NOW: call `weights.__dict__.pop("_fp8_layers", None)` after loading others.)"""
    cache = weights.__dict__.get("_fp8_layers")
    if cache is None:
        cache = [{n: ops.quantize_weight_fp8(ops.unpack_weight(getattr(lw, n))) for n in ("qkv_w", "o_w", "gu_w", "down_w")}
                 for lw in weights.layers]
        weights.__dict__["_fp8_layers"] = cache
    return cache


class OpenVLAEngine:
    def __init__(self, weights: VLAWeights, batch: int, prompt_len: int, n_new: int = 7, all_rows: bool = False,
                 use_mask: bool = False, splitk: bool = False, fp8: bool = False, padded: bool = False,
                 vision_only: bool = False, text_only: bool = False):
        """all_rows=True builds the training/eval-style forward instead of generation: logits for every position
        (`logits_all` [B*S, vocab] fp32) and no decode steps. use_mask=True threads a [B, S] uint8 key-padding mask
        (`key_mask`, 1 = attend) through the Llama attention (modeling_prismatic.py:387-390). splitk=True lets
        bl_gemm_bf16 split the K range of the last, partially filled round of tiles (≈ +2 % throughput at 7B) — OFF by
        default because those rows then sum in a different fp32 order than the rest, so a sequence's result would depend
        on its batch slot (it breaks "batch-B ≡ B × batch-1 bit for bit", tests/test_full_size_gpu.py). fp8=True runs the
        Llama prefill projections (qkv, o, gate/up, down of every layer but the last) as W8A8 e4m3 GEMMs on the
        block-scaled MFMA (BASELINE configs[4]; per-token × per-channel scales, ops.gemm_fp8) — an extension with no
        reference counterpart: results agree with the bf16 path to quantisation noise, not bit for bit. padded=True builds
        the generation plan for a batch of RIGHT-PADDED prompts (`set_padded_inputs`; HF generation with an attention
        mask, modeling_prismatic.py:387-390): pad positions are hidden from every attention, each sequence's first token
        comes from ITS last real position, and the new tokens are rotated at the sequence's own position — every
        sequence gets exactly the ids and logits it gets alone, un-padded (tests/test_hf_boundary_gpu.py).
        vision_only=True plans the towers alone (the training step's frozen front end): no Llama plans, so nothing here
        holds the decoder-layer weights (parameter-sharded training frees them). text_only=True plans the reference's
        language-only forward (`pixel_values is None`, modeling_prismatic.py:343-359): no towers, no projector, S = L."""
        self.w, self.dims = weights, weights.dims
        self.vision_only, self.text_only = vision_only, text_only
        if text_only and (vision_only or padded or fp8):
            raise ValueError("text_only is the plain bf16 language-model plan")
        self.epoch = 0            # bumped by every prefill that overwrites the KV caches (cache handles check it)
        if not vision_only and not weights.layers_resident:
            raise RuntimeError("the decoder-layer weights are sharded out of the model (parameter-sharded training in progress): "
                               "call the strategy's finish() / TrainStep.materialize_params() before building an inference engine")
        self.padded = padded
        use_mask = use_mask or padded
        if padded and (all_rows or fp8):
            raise ValueError("padded generation is built for the bf16 generation plan")
        d = self.dims
        if all_rows:
            n_new = 1
        self.all_rows, self.use_mask = all_rows, use_mask
        self.B, self.L, self.n_new = batch, prompt_len, n_new
        self.n_patches = 0 if text_only else d.n_patches
        self.S = prompt_len + self.n_patches
        self.cache_len = (self.S + n_new + 63) // 64 * 64
        if self.cache_len > d.max_pos:
            raise ValueError("sequence exceeds max_position_embeddings")
        dev = weights.embed.device
        self.device = dev
        B, S, D, I = batch, self.S, d.llm_dim, d.llm_inter
        z = lambda *shape, dtype=torch.bfloat16: torch.zeros(*shape, dtype=dtype, device=dev)
        # inputs / outputs (static addresses so a captured graph can be replayed)
        self.pixel_values = z(B, 6, 224, 224)
        self.input_ids = z(B, prompt_len, dtype=torch.int64)
        self.gen_ids = z(n_new, B, dtype=torch.int64)
        self.logits = z(n_new, B, d.vocab, dtype=torch.float32)
        # vision buffers (sized for the larger tower, shared by both: they run back to back on one stream)
        tmax = max(d.dino.tokens, d.siglip.tokens)
        dmax = max(d.dino.dim, d.siglip.dim)
        hmax = max(d.dino.mlp_pad, d.siglip.mlp_pad)
        # one private set per tower: the two towers run concurrently on two streams
        self.vbuf = [dict(col=z(B * 256, (d.patch_k + 63) // 64 * 64), x=z(B * tmax * dmax), h=z(B * tmax * dmax),
                          ao=z(B * tmax * dmax), qkv=z(B * tmax * 3 * dmax), mlp=z(B * tmax * hmax),
                          ws=None) for _ in range(2)]   # ViT GEMMs never reach the split-K regime (K <= 4352)
        self.feats = z(B * 256, d.vision_dim)
        self.p1, self.p2 = z(B * 256, 4 * d.vision_dim), z(B * 256, D)
        # llm buffers
        if not vision_only:
            self.x = z(B, S, D)
            self.h, self.ao = z(B * S, D), z(B * S, D)
            self.qkv = z(B * S, 3 * D)
            self.act = z(B * S, I)
            self.k_cache = [z(B, d.llm_heads, self.cache_len, d.head_dim) for _ in range(d.llm_layers)]
            self.v_cache = [z(B, d.llm_heads, self.cache_len, d.head_dim) for _ in range(d.llm_layers)]
            self.xd, self.hd, self.aod = z(B, D), z(B, D), z(B, D)
            self.qkvd, self.actd = z(B, 3 * D), z(B, I)
        self.cos, self.sin = rope_tables(d.head_dim, d.max_pos, d.rope_theta, dev)
        self.ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev) if splitk else None   # split-K scratch (opt-in)
        if padded:        # one mask over the whole cache: real prompt rows and every generated row are visible
            if d.head_dim != 128 or not ops.skinny_supported(B, D, EPI_NONE):
                raise NotImplementedError("padded generation needs head_dim 128 and a batch of at most 16")
            self.cache_mask = torch.ones(B, self.cache_len, dtype=torch.uint8, device=dev)
            self.key_mask = self.cache_mask[:, :S]
            self.last_row = torch.full((B,), S - 1, dtype=torch.int64, device=dev)      # index of the last real position
            self.rope_pos = torch.zeros(n_new, B, dtype=torch.int32, device=dev)        # rotation position of new token t
            self.q_last, self.x_last = z(B, 3 * D), z(B, D)
            self._rows = torch.arange(B, device=dev)
        else:
            self.key_mask = torch.ones(B, S, dtype=torch.uint8, device=dev) if use_mask else None
        self.logits_all = z(B * S, d.vocab, dtype=torch.float32) if all_rows else None
        self.fp8 = fp8
        if fp8:
            if d.llm_dim % 128 or d.llm_inter % 128:
                raise ValueError("fp8 prefill needs llm_dim and llm_inter to be multiples of 128")
            self.w8 = _fp8_layers(weights)
            self.h8, self.act8 = z(B * S, D, dtype=torch.uint8), z(B * S, I, dtype=torch.uint8)
            self.sq = z(B * S, dtype=torch.float32)

        self.dino_ops = [] if text_only else self._plan_tower(weights.dino, 0, self.vbuf[0])
        self.siglip_ops = [] if text_only else self._plan_tower(weights.siglip, d.dino.dim, self.vbuf[1])
        self.vision_ops = self.dino_ops + self.siglip_ops      # serial order (profiling / single-stream use)
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        if vision_only:
            self.projector_ops, self.prefill_ops, self.decode_ops = [], [], []
        else:
            self.projector_ops = [] if text_only else self._plan_projector()
            self.prefill_ops = self._plan_prefill()
            self.decode_ops = [self._plan_decode(t) for t in range(1, n_new)]
        self._graph: Optional[torch.cuda.CUDAGraph] = None

    def _g(self, *args, **kw):
        return ops.gemm(*args, workspace=self.ws, **kw)

    # ---- plans ----------------------------------------------------------------------------------------------------
    def _plan_tower(self, tw: TowerW, feat_col: int, vb: dict) -> List[Op]:
        """timm VisionTransformer up to the tap: K1-K9 of SURVEY §2.4."""
        t, B = tw.dims, self.B
        T, Dm, Hp, hd = t.tokens, t.dim, t.mlp_pad, t.head_dim
        M = B * T
        x = vb["x"][:M * Dm].view(M, Dm)
        h = vb["h"][:M * Dm].view(M, Dm)
        ao = vb["ao"][:M * Dm].view(M, Dm)
        qkv = vb["qkv"][:M * 3 * Dm].view(M, 3 * Dm)
        mlp = vb["mlp"][:M * Hp].view(M, Hp)
        v_col = vb["col"]
        g = lambda *a, **k: ops.gemm(*a, workspace=vb["ws"], **k)     # private split-K scratch per stream
        plan = [ops.im2col_patch14(self.pixel_values, t.chan0, v_col, run=False)]
        if tw.prefix is not None:
            plan.append(ops.write_prefix_tokens(tw.prefix, x, B, T, run=False))
        # patch-embed GEMM + bias + pos-embed, rows written behind the prefix tokens
        plan.append(g(v_col, tw.patch_w, x, EPI_BIAS_RES, bias=tw.patch_b, res=tw.pos, res_row_mod=256,
                             out_map=(256, T, t.n_prefix), algo_nk=(Dm, self.dims.patch_k), run=False))
        st = (T * 3 * Dm, hd, 3 * Dm)
        for i, b in enumerate(tw.blocks):
            plan.append(ops.layernorm(x, b.norm1_w, b.norm1_b, h, self.dims.ln_eps, run=False))
            plan.append(g(h, b.qkv_w, qkv, EPI_BIAS, bias=b.qkv_b, run=False))
            plan.append(ops.attention(qkv, qkv[:, Dm:], qkv[:, 2 * Dm:], ao, B=B, H=t.heads, Sq=T, Skv=T, head_dim=hd,
                                      q_strides=st, k_strides=st, v_strides=st, o_strides=(T * Dm, hd, Dm),
                                      causal=False, run=False))
            plan.append(g(ao, b.proj_w, x, EPI_BIAS_RES, bias=b.proj_b, scale=b.ls1, res=x, run=False))
            plan.append(ops.layernorm(x, b.norm2_w, b.norm2_b, h, self.dims.ln_eps, run=False))
            plan.append(g(h, b.fc1_w, mlp, EPI_BIAS_GELU, bias=b.fc1_b, algo_nk=(t.mlp, Dm), run=False))
            if i + 1 < len(tw.blocks):
                plan.append(g(mlp, b.fc2_w, x, EPI_BIAS_RES, bias=b.fc2_b, scale=b.ls2, res=x, algo_nk=(Dm, t.mlp),
                                     run=False))
            else:   # tap: drop the prefix tokens and write this tower's channels of the fused feature map
                plan.append(g(mlp, b.fc2_w, self.feats[:, feat_col:feat_col + Dm], EPI_BIAS_RES, bias=b.fc2_b,
                                     scale=b.ls2, res=x, out_map=(T, 256, -t.n_prefix), algo_nk=(Dm, t.mlp), run=False))
        return plan

    def _plan_projector(self) -> List[Op]:
        """PrismaticProjector (modeling_prismatic.py:151-156); fc3 writes straight into LLM embedding rows 1..256."""
        w, B, S, D = self.w, self.B, self.S, self.dims.llm_dim
        return [self._g(self.feats, w.fc1_w, self.p1, EPI_BIAS_GELU, bias=w.fc1_b, run=False),
                self._g(self.p1, w.fc2_w, self.p2, EPI_BIAS_GELU, bias=w.fc2_b, run=False),
                self._g(self.p2, w.fc3_w, self.x.view(B * S, D), EPI_BIAS, bias=w.fc3_b, out_map=(256, S, 1), run=False)]

    def _plan_prefill(self, b0: int = 0, b1: Optional[int] = None) -> List[Op]:
        """Llama prefill + first token for batch rows [b0, b1) (default: the whole batch). Every buffer is batch-major, so
        a batch range is a contiguous slice of each; results do not depend on the range a sequence is run in."""
        d, w, S = self.dims, self.w, self.S
        b1 = self.B if b1 is None else b1
        B = b1 - b0
        D, H, hd = d.llm_dim, d.llm_heads, d.head_dim
        rows = slice(b0 * S, b1 * S)
        x3 = self.x[b0:b1]
        x = x3.view(B * S, D)
        h, qkv, ao, act = self.h[rows], self.qkv[rows], self.ao[rows], self.act[rows]
        xd, hdd, aod, actd = self.xd[b0:b1], self.hd[b0:b1], self.aod[b0:b1], self.actd[b0:b1]
        key_mask = self.key_mask[b0:b1] if self.key_mask is not None else None
        plan = [ops.embed_splice(self.input_ids[b0:b1], w.embed, x3, self.n_patches, run=False)]
        self.layer_ends: List[int] = []       # len(plan) after each decoder layer (hidden-state taps of the all_rows plan)
        cs = (H * self.cache_len * hd, self.cache_len * hd, hd)
        # Generation consumes only the last position of the last layer (the reference materialises all S rows of every
        # layer): that layer still projects K/V for every position (the decode steps attend to them),
        # but its attention, o_proj, MLP run on the B last rows only — single-query attention + weight-streaming GEMMs.
        last_rows_only = not self.all_rows and d.llm_layers > 1
        for l, lw in enumerate(w.layers):
            kc, vc = self.k_cache[l][b0:b1], self.v_cache[l][b0:b1]
            plan.append(ops.rmsnorm(x, lw.ln1, h, d.rms_eps, run=False))
            plan.append(self._g(h, lw.qkv_w, qkv, EPI_NONE, run=False))
            last = last_rows_only and l == d.llm_layers - 1
            if self.fp8 and not last and hd == 128 and S <= 320:
                w8, h8, act8, sq = self.w8[l], self.h8[rows], self.act8[rows], self.sq[rows]
                q8 = lambda src, dst: ops.quantize_rows_fp8(src, dst, sq, run=False)[2]
                plan[-1] = q8(h, h8)                                           # replaces the bf16 qkv GEMM appended above
                plan.append(ops.gemm_fp8(h8, sq, *w8["qkv_w"], qkv, EPI_NONE, run=False))
                plan.append(ops.attention_rope(qkv, kc, vc, ao, self.cos, self.sin, B=B, S=S, H=H, head_dim=hd, pos0=0,
                                               key_mask=key_mask, run=False))
                plan.append(q8(ao, h8))
                plan.append(ops.gemm_fp8(h8, sq, *w8["o_w"], x, EPI_RES, res=x, run=False))
                plan.append(ops.rmsnorm(x, lw.ln2, h, d.rms_eps, run=False))
                plan.append(q8(h, h8))
                plan.append(ops.gemm_fp8(h8, sq, *w8["gu_w"], act, EPI_SWIGLU, run=False))
                plan.append(q8(act, act8))
                plan.append(ops.gemm_fp8(act8, sq, *w8["down_w"], x, EPI_RES, res=x, run=False))
                self.layer_ends.append(len(plan))
                continue
            if not last and hd == 128 and S <= 320:
                # RoPE and the KV-cache write ride inside the attention kernel's q / k / v loads
                plan.append(ops.attention_rope(qkv, kc, vc, ao, self.cos, self.sin, B=B, S=S, H=H, head_dim=hd, pos0=0,
                                               key_mask=key_mask, run=False))
                plan.append(self._g(ao, lw.o_w, x, EPI_RES, res=x, run=False))
                plan.append(ops.rmsnorm(x, lw.ln2, h, d.rms_eps, run=False))
                plan.append(self._g(h, lw.gu_w, act, EPI_SWIGLU, run=False))
                plan.append(self._g(act, lw.down_w, x, EPI_RES, res=x, run=False))
                self.layer_ends.append(len(plan))
                continue
            plan.append(ops.rope_kvcache(qkv, self.cos, self.sin, kc, vc, B=B, S=S, H=H, head_dim=hd, pos0=0, run=False))
            if last and self.padded:
                # each sequence's last REAL position: gather its (rotated) q row and residual row (index glue, no
                # arithmetic), then the same single-query attention / weight-streaming GEMMs as the un-padded plan
                def gather(qkv=qkv, x3=x3):
                    self.q_last.copy_(qkv.view(B, S, 3 * D)[self._rows, self.last_row])
                    self.x_last.copy_(x3[self._rows, self.last_row])
                plan.append(ops.glue("gather_last_real_rows", gather, (qkv, x3)))
                q_last, x_last = self.q_last, self.x_last
                plan.append(ops.attention_decode(q_last, kc, vc, aod, B=B, H=H, Skv=S, head_dim=hd,
                                                 q_strides=(3 * D, hd, 3 * D), k_strides=cs, v_strides=cs,
                                                 o_strides=(D, hd, D), key_mask=self.cache_mask, run=False))
            elif last:
                q_last = qkv.view(B, S, 3 * D)[:, S - 1]                       # roped in place; row stride S·3D
                x_last = x3[:, S - 1, :]
                plan.append(ops.attention_decode(q_last, kc, vc, aod, B=B, H=H, Skv=S, head_dim=hd,
                                                 q_strides=(S * 3 * D, hd, 3 * D), k_strides=cs, v_strides=cs,
                                                 o_strides=(D, hd, D), run=False))
            if last:
                plan.append(self._g(aod, lw.o_w, xd, EPI_RES, res=x_last, run=False))
                if ops.skinny_supported(B, D, EPI_SWIGLU):
                    plan.append(self._g(xd, lw.gu_w, actd, EPI_SWIGLU, a_norm=(lw.ln2, d.rms_eps), run=False))
                else:
                    plan.append(ops.rmsnorm(xd, lw.ln2, hdd, d.rms_eps, run=False))
                    plan.append(self._g(hdd, lw.gu_w, actd, EPI_SWIGLU, run=False))
                plan.append(self._g(actd, lw.down_w, xd, EPI_RES, res=xd, run=False))
                return plan + self._head(xd, 0, b0, b1)
            plan.append(ops.attention(qkv, kc, vc, ao, B=B, H=H, Sq=S, Skv=S, head_dim=hd,
                                      q_strides=(S * 3 * D, hd, 3 * D), k_strides=cs, v_strides=cs,
                                      o_strides=(S * D, hd, D), causal=True, key_mask=key_mask, run=False))
            plan.append(self._g(ao, lw.o_w, x, EPI_RES, res=x, run=False))
            plan.append(ops.rmsnorm(x, lw.ln2, h, d.rms_eps, run=False))
            plan.append(self._g(h, lw.gu_w, act, EPI_SWIGLU, run=False))
            plan.append(self._g(act, lw.down_w, x, EPI_RES, res=x, run=False))
            self.layer_ends.append(len(plan))
        if self.all_rows:   # HF semantics: logits for every position (what forward()/training consume)
            plan.append(ops.rmsnorm(x, w.norm, h, d.rms_eps, run=False))
            plan.append(self._g(h, w.lm_head, self.logits_all[rows], EPI_F32_BF16R, run=False))
            return plan
        # final norm + lm_head on the last position only (the reference materialises all S rows)
        plan += self._head(x3[:, S - 1, :], 0, b0, b1)
        return plan

    def _head(self, x_rows: torch.Tensor, t: int, b0: int = 0, b1: Optional[int] = None) -> List[Op]:
        """final RMSNorm → lm_head (bf16-rounded fp32 logits) → greedy argmax, for generation step t."""
        d, w = self.dims, self.w
        b1 = self.B if b1 is None else b1
        logits, ids, hdd = self.logits[t][b0:b1], self.gen_ids[t][b0:b1], self.hd[b0:b1]
        if ops.skinny_supported(b1 - b0, d.llm_dim, EPI_F32_BF16R):   # RMSNorm fused into the weight-streaming GEMM
            plan = [self._g(x_rows, w.lm_head, logits, EPI_F32_BF16R, a_norm=(w.norm, d.rms_eps), run=False)]
        else:
            plan = [ops.rmsnorm(x_rows, w.norm, hdd, d.rms_eps, run=False),
                    self._g(hdd, w.lm_head, logits, EPI_F32_BF16R, run=False)]
        plan.append(ops.argmax(logits, ids, run=False))
        return plan

    def _plan_decode(self, t: int) -> List[Op]:
        """Cached-generation step t (modeling_prismatic.py:325-341): token gen_ids[t-1] at position S+t-1. Five
        launches per layer: qkv (RMSNorm fused) → attention (RoPE + KV append fused) → o_proj+residual → gate/up
        (RMSNorm + SwiGLU fused) → down+residual."""
        d, w, B = self.dims, self.w, self.B
        D, H, hd = d.llm_dim, d.llm_heads, d.head_dim
        pos = self.S + t - 1
        fused = ops.skinny_supported(B, D, EPI_NONE) and hd == 128
        cs = (H * self.cache_len * hd, self.cache_len * hd, hd)
        plan = [ops.embed_splice(self.gen_ids[t - 1].view(B, 1), w.embed, self.xd.view(B, 1, D), 0, run=False)]
        for l, lw in enumerate(w.layers):
            if fused:
                plan.append(self._g(self.xd, lw.qkv_w, self.qkvd, EPI_NONE, a_norm=(lw.ln1, d.rms_eps), run=False))
                # padded: every sequence appends at its own position (over its pad rows): the cache layout, and with
                # it every sum, is the one of the sequence's un-padded run
                pad_kw = dict(rope_pos=self.rope_pos[t]) if self.padded else {}
                plan.append(ops.attention_decode_rope(self.qkvd, self.k_cache[l], self.v_cache[l], self.aod, self.cos,
                                                      self.sin, B=B, H=H, head_dim=hd, pos=pos, run=False, **pad_kw))
            else:
                plan.append(ops.rmsnorm(self.xd, lw.ln1, self.hd, d.rms_eps, run=False))
                plan.append(self._g(self.hd, lw.qkv_w, self.qkvd, EPI_NONE, run=False))
                plan.append(ops.rope_kvcache(self.qkvd, self.cos, self.sin, self.k_cache[l], self.v_cache[l], B=B, S=1,
                                             H=H, head_dim=hd, pos0=pos, run=False))
                plan.append(ops.attention_decode(self.qkvd, self.k_cache[l], self.v_cache[l], self.aod, B=B, H=H,
                                                 Skv=pos + 1, head_dim=hd, q_strides=(3 * D, hd, 3 * D), k_strides=cs,
                                                 v_strides=cs, o_strides=(D, hd, D), run=False))
            plan.append(self._g(self.aod, lw.o_w, self.xd, EPI_RES, res=self.xd, run=False))
            if fused:
                plan.append(self._g(self.xd, lw.gu_w, self.actd, EPI_SWIGLU, a_norm=(lw.ln2, d.rms_eps), run=False))
            else:
                plan.append(ops.rmsnorm(self.xd, lw.ln2, self.hd, d.rms_eps, run=False))
                plan.append(self._g(self.hd, lw.gu_w, self.actd, EPI_SWIGLU, run=False))
            plan.append(self._g(self.actd, lw.down_w, self.xd, EPI_RES, res=self.xd, run=False))
        plan += self._head(self.xd, t)
        return plan

    # ---- execution ------------------------------------------------------------------------------------------------
    def all_ops(self) -> List[Op]:
        out = self.vision_ops + self.projector_ops + self.prefill_ops
        for step in self.decode_ops:
            out = out + step
        return out

    def run_vision(self) -> None:
        """The two towers are independent until the projector: DINOv2 on the current stream, SigLIP on a side stream
        (fork/join with stream waits, which HIP-graph capture records as parallel branches). Their GEMMs often fill only
        part of the chip (e.g. 264 tiles on 512 slots), so running them side by side recovers the idle CUs."""
        if self._side is None:
            ops.run_all(self.vision_ops)
            return
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            ops.run_all(self.siglip_ops)
        ops.run_all(self.dino_ops)
        main.wait_stream(self._side)

    def _run_all_stages(self) -> None:
        self.run_vision()
        ops.run_all(self.projector_ops + self.prefill_ops)
        for step in self.decode_ops:
            ops.run_all(step)

    def run_eager(self) -> None:
        self._run_all_stages()

    def capture(self) -> None:
        """Capture the whole action-sequence computation into one HIP graph (after one eager warm-up launch)."""
        self.run_eager()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._run_all_stages()
        self._graph = g

    def replay(self) -> None:
        if self._graph is None:
            self.run_eager()
        else:
            self._graph.replay()

    def set_inputs(self, input_ids: torch.Tensor, pixel_values: torch.Tensor) -> None:
        if tuple(input_ids.shape) != (self.B, self.L):
            raise ValueError(f"engine built for input_ids {(self.B, self.L)}, got {tuple(input_ids.shape)}")
        if self.text_only:
            if pixel_values is not None:
                raise ValueError("a text_only engine takes no pixel_values")
            self.input_ids.copy_(input_ids)
            self.epoch += 1
            return
        if tuple(pixel_values.shape) != (self.B, 6, 224, 224):
            raise ValueError(f"pixel_values must be [{self.B}, 6, 224, 224]")
        self.input_ids.copy_(input_ids)
        self.pixel_values.copy_(pixel_values.to(torch.bfloat16))
        self.epoch += 1

    def set_padded_inputs(self, input_ids: torch.Tensor, pixel_values: torch.Tensor, attention_mask: torch.Tensor) -> None:
        """Right-padded prompts [B, L] with attention_mask [B, L] (1 = real token; the collator's layout,
        util/data_utils.py:101-142). Fills the cache mask, the per-sequence last position and rotation positions."""
        if not self.padded:
            raise ValueError("engine was not built with padded=True")
        m = attention_mask.to(self.device).bool()
        if tuple(m.shape) != (self.B, self.L):
            raise ValueError(f"attention_mask must be {(self.B, self.L)}")
        n_real = m.sum(dim=1)
        if bool((n_real < 1).any()) or not bool((m == (torch.arange(self.L, device=self.device)[None, :] < n_real[:, None])).all()):
            raise ValueError("generate(): prompts must be right-padded (attention_mask = 1…1 0…0) with at least one token")
        self.set_inputs(input_ids, pixel_values)
        P = self.dims.n_patches
        self.cache_mask.fill_(1)
        self.cache_mask[:, 1 + P:self.S] = m[:, 1:].to(torch.uint8)          # column 0 (BOS) and the 256 patch columns stay on
        self.last_row.copy_(P + n_real - 1)
        self.rope_pos.copy_((P + n_real)[None, :].to(torch.int32) + torch.arange(self.n_new, device=self.device, dtype=torch.int32)[:, None] - 1)

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, pixel_values: torch.Tensor) -> torch.Tensor:
        """Greedy n_new tokens for every sequence. Returns int64 [B, n_new] (device tensor; no host sync inside)."""
        self.set_inputs(input_ids, pixel_values)
        self.replay()
        return self.gen_ids.t()
