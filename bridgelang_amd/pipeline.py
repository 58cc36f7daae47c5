"""Software pipelines over batches for throughput serving: `TwoStagePipeline` (two batches in flight) and
`StaggeredDecodePipeline` (n_new batches in flight, their decode iterations merged into one weight pass per step).


A `predict_action` batch has two very different halves: vision + prefill is MFMA-bound (≈ 65 ms at B = 16) but leaves
CUs idle in partial GEMM rounds and between dependent launches; the 6 cached decode steps are HBM-bound weight streaming
(≈ 27 ms). `TwoStagePipeline` keeps two engines (two sets of activations / KV caches over ONE set of weights) and, per
step, runs stage 1 (vision → projector → prefill → first token) of the batch submitted NOW on one stream while stage 2
(decode steps 1..6) of the batch submitted ONE STEP EARLIER runs on a second stream; the step completes one batch.
Work per step is exactly one batch's full computation; only the latency of an individual batch is two steps.
Each of the two stage pairings is captured once as a HIP graph with fork/join stream dependencies.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops
from .engine import OpenVLAEngine
from .ops import EPI_F32_BF16R, EPI_NONE, EPI_RES, EPI_SWIGLU, Op
from .weights import VLAWeights


class TwoStagePipeline:
    def __init__(self, weights: VLAWeights, batch: int, prompt_len: int, n_new: int = 7):
        self.engines = [OpenVLAEngine(weights, batch, prompt_len, n_new) for _ in range(2)]
        self.device = weights.embed.device
        self._decode_stream = torch.cuda.Stream(device=self.device)
        self._graphs: List[Optional[torch.cuda.CUDAGraph]] = [None, None]
        self._tick = 0

    def _run_pair(self, k: int) -> None:
        """stage 1 of engine k ‖ stage 2 of engine 1-k."""
        e1, e2 = self.engines[k], self.engines[1 - k]
        main = torch.cuda.current_stream()
        self._decode_stream.wait_stream(main)
        with torch.cuda.stream(self._decode_stream):
            for step in e2.decode_ops:
                ops.run_all(step)
        e1.run_vision()
        ops.run_all(e1.projector_ops + e1.prefill_ops)
        main.wait_stream(self._decode_stream)

    def capture(self) -> None:
        for k in (0, 1):
            self._run_pair(k)          # eager warm-up (also sets kernel attributes outside capture)
        torch.cuda.synchronize()
        for k in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._run_pair(k)
            self._graphs[k] = g

    @torch.no_grad()
    def step(self, input_ids: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Submit a batch (or re-use the inputs already resident in the engine's buffers when None) and return the
        [B, n_new] token ids of the batch submitted one step earlier (garbage on the very first call)."""
        k = self._tick & 1
        if input_ids is not None:
            self.engines[k].set_inputs(input_ids, pixel_values)
        if self._graphs[k] is not None:
            self._graphs[k].replay()
        else:
            self._run_pair(k)
        self._tick += 1
        return self.engines[1 - k].gen_ids.t()

    def flush(self) -> torch.Tensor:
        """Finish the batch still in stage 1 (drain the pipeline); returns its ids."""
        k = (self._tick - 1) & 1
        for step in self.engines[k].decode_ops:
            ops.run_all(step)
        return self.engines[k].gen_ids.t()


class StaggeredDecodePipeline:
    """n_new batches in flight, one submitted and one completed per step (continuous batching for a fixed-length decode).

    A cached decode iteration streams all 13.2 GB of Llama weights for B = 16 rows, six times per batch. With one batch
    submitted per step, the batches submitted 1, 2, … n_new-1 steps ago are exactly at decode iterations 1, 2, … n_new-1:
    their rows are stacked into ONE (n_new-1)·B-row iteration, so every weight matrix is streamed once per step instead of
    n_new-1 times (the rows go through the tiled GEMMs; only attention — own KV cache, own position — runs per batch).
    Per step: vision + projector + prefill of the new batch on one stream ‖ the merged decode iteration on another; the
    work per step is exactly one batch's full computation, a batch's latency is n_new steps. Slot s of the n_new slots
    (KV caches, ids, prefill activations = one OpenVLAEngine each) holds the batch submitted at step ≡ s (mod n_new); the
    n_new slot rotations are captured as n_new HIP graphs.

    Results per sequence equal the plain engine's bit for bit: the stacked rows go through kernels that reproduce the
    per-batch kernels' fp32 summation order (_plan_merged; tests/test_pipeline_gpu.py).
    """

    def __init__(self, weights: VLAWeights, batch: int, prompt_len: int, n_new: int = 7, split_vision: bool = False,
                 fp8: bool = False):
        """split_vision=True adds a third stage: the vision towers + projector of the batch submitted NOW run beside the
        Llama prefill of the batch submitted one step earlier (n_new + 1 slots, latency n_new + 1 steps)."""
        if n_new < 2:
            raise ValueError("StaggeredDecodePipeline needs at least one decode iteration (n_new >= 2)")
        self.w, self.dims, self.B, self.n_new = weights, weights.dims, batch, n_new
        self.split_vision = split_vision
        self.lag = 1 if split_vision else 0         # steps between a batch's submission and its prefill
        self.slots = n_new + self.lag
        self.engines = [OpenVLAEngine(weights, batch, prompt_len, n_new, fp8=fp8) for _ in range(self.slots)]
        self.device = dev = weights.embed.device
        d = self.dims
        G = n_new - 1
        M = G * batch
        z = lambda *shape, dtype=torch.bfloat16: torch.zeros(*shape, dtype=dtype, device=dev)
        self.xd, self.hd, self.aod = z(M, d.llm_dim), z(M, d.llm_dim), z(M, d.llm_dim)
        self.qkvd, self.actd = z(M, 3 * d.llm_dim), z(M, d.llm_inter)
        self.logits = z(M, d.vocab, dtype=torch.float32)       # rows (g-1)·B … g·B: decode iteration g of this step
        # fp32 partials of the merged decode GEMMs of the narrow layers (N = llm_dim: four workgroups share K — an exact
        # split of the skinny order's binary tree, see bl_gemm_skinny_rows_bf16)
        self.ws = torch.empty(4 * min(M, 128) * d.llm_dim * 4, dtype=torch.uint8, device=dev)
        self._decode_stream = torch.cuda.Stream(device=dev)
        self._vision_stream = torch.cuda.Stream(device=dev)
        self._main_stream = torch.cuda.Stream(device=dev)    # capture stream (stream priorities were tried: no effect)
        self.merged_ops: List[List[Op]] = [self._plan_merged(k) for k in range(self.slots)]
        self._graphs: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.slots
        self._tick = 0

    def _plan_merged(self, k: int) -> List[Op]:
        """Decode iteration g = 1 … n_new-1 of the batch in slot (k - g) mod n_new, all in one pass over the weights
        (the per-batch steps of OpenVLAEngine._plan_decode, modeling_prismatic.py:325-341, stacked on the row axis).

        Every op mirrors the kernel choice OpenVLAEngine._plan_decode makes for ONE batch, in its many-rows form with the
        same arithmetic: where the engine streams weights through bl_gemm_skinny_bf16 (B <= 16), the stacked rows go
        through bl_gemm_skinny_rows_bf16 (same 8-way K partition and combine order) and the fused a_norm becomes
        bl_rmsnorm_skinny_bf16; where the engine falls back to the tiled kernels, so does this plan (no split-K workspace:
        the tiled kernels' K order does not depend on the row count). Hence ids and logits equal the plain engine's bit
        for bit (tests/test_pipeline_gpu.py, tests/test_full_size_gpu.py)."""
        d, w, B = self.dims, self.w, self.B
        D, H, hd = d.llm_dim, d.llm_heads, d.head_dim
        e0 = self.engines[0]
        fused = ops.skinny_supported(B, D, EPI_NONE) and hd == 128       # OpenVLAEngine._plan_decode's condition
        M = self.xd.shape[0]

        def gm(A, W, out, epi, res=None):
            """one Linear over the stacked rows, in the arithmetic the engine uses for B rows"""
            if not ops.skinny_supported(B, A.shape[1], epi):
                return [ops.gemm(A, W, out, epi, res=res, skinny=False, run=False)]
            return [ops.gemm(A[r0:r0 + 128], W, out[r0:r0 + 128], epi, res=None if res is None else res[r0:r0 + 128],
                             skinny_rows=True, workspace=self.ws, run=False) for r0 in range(0, M, 128)]

        def norm(x, wn, out, fused_in_engine):
            return (ops.rmsnorm_skinny if fused_in_engine else ops.rmsnorm)(x, wn, out, d.rms_eps, run=False)

        groups = [(g, self.engines[(k - g - self.lag) % self.slots], slice((g - 1) * B, g * B)) for g in range(1, self.n_new)]
        plan = [ops.embed_splice(e.gen_ids[g - 1].view(B, 1), w.embed, self.xd[r].view(B, 1, D), 0, run=False)
                for g, e, r in groups]
        for l, lw in enumerate(w.layers):
            plan.append(norm(self.xd, lw.ln1, self.hd, fused))
            plan += gm(self.hd, lw.qkv_w, self.qkvd, EPI_NONE)
            grouped = fused and len(groups) <= 8
            if grouped:     # all decode iterations' attention in one launch
                plan.append(ops.attention_decode_rope_grouped(
                    self.qkvd, [e.k_cache[l] for _, e, _ in groups], [e.v_cache[l] for _, e, _ in groups], self.aod,
                    e0.cos, e0.sin, B=B, H=H, head_dim=hd, pos=[e.S + g - 1 for g, e, _ in groups], run=False))
            for g, e, r in groups:
                pos = e.S + g - 1
                if grouped:
                    continue
                if fused:
                    plan.append(ops.attention_decode_rope(self.qkvd[r], e.k_cache[l], e.v_cache[l], self.aod[r], e0.cos, e0.sin,
                                                          B=B, H=H, head_dim=hd, pos=pos, run=False))
                else:
                    cs = (H * e.cache_len * hd, e.cache_len * hd, hd)
                    plan.append(ops.rope_kvcache(self.qkvd[r], e0.cos, e0.sin, e.k_cache[l], e.v_cache[l], B=B, S=1, H=H,
                                                 head_dim=hd, pos0=pos, run=False))
                    plan.append(ops.attention_decode(self.qkvd[r], e.k_cache[l], e.v_cache[l], self.aod[r], B=B, H=H,
                                                     Skv=pos + 1, head_dim=hd, q_strides=(3 * D, hd, 3 * D), k_strides=cs,
                                                     v_strides=cs, o_strides=(D, hd, D), run=False))
            plan += gm(self.aod, lw.o_w, self.xd, EPI_RES, res=self.xd)
            plan.append(norm(self.xd, lw.ln2, self.hd, fused))
            plan += gm(self.hd, lw.gu_w, self.actd, EPI_SWIGLU)
            plan += gm(self.actd, lw.down_w, self.xd, EPI_RES, res=self.xd)
        plan.append(norm(self.xd, w.norm, self.hd, ops.skinny_supported(B, D, EPI_F32_BF16R)))   # OpenVLAEngine._head
        plan += gm(self.hd, w.lm_head, self.logits, EPI_F32_BF16R)
        plan += [ops.argmax(self.logits[r], e.gen_ids[g], run=False) for g, e, r in groups]
        return plan

    def _run_tick(self, k: int) -> None:
        """stage 1 of slot k ‖ the merged decode iteration over the other slots."""
        e1 = self.engines[k]
        main = torch.cuda.current_stream()
        self._decode_stream.wait_stream(main)
        with torch.cuda.stream(self._decode_stream):
            ops.run_all(self.merged_ops[k])
        if self.split_vision:
            vs, ss = self._vision_stream, e1._side      # both towers fork from the main stream (no nested fork)
            vs.wait_stream(main)
            ss.wait_stream(main)
            with torch.cuda.stream(ss):
                ops.run_all(e1.siglip_ops)
            with torch.cuda.stream(vs):
                ops.run_all(e1.dino_ops)
                vs.wait_stream(ss)
                ops.run_all(e1.projector_ops)
            ops.run_all(self.engines[(k - 1) % self.slots].prefill_ops)
            main.wait_stream(vs)
        else:
            e1.run_vision()
            ops.run_all(e1.projector_ops + e1.prefill_ops)
        main.wait_stream(self._decode_stream)

    def capture(self) -> None:
        for k in range(self.slots):
            self._run_tick(k)          # eager warm-up (also sets kernel attributes outside capture)
        torch.cuda.synchronize()
        for k in range(self.slots):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._main_stream):
                self._run_tick(k)
            self._graphs[k] = g

    @torch.no_grad()
    def step(self, input_ids: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Submit a batch (None: re-use the inputs resident in the slot's buffers) and return the [B, n_new] ids of the
        batch submitted n_new-1 steps earlier, which this step completed (garbage until the pipeline has filled). The
        returned view is overwritten by the next step() — copy it first."""
        k = self._tick % self.slots
        if input_ids is not None:
            self.engines[k].set_inputs(input_ids, pixel_values)
        if self._graphs[k] is not None:
            self._graphs[k].replay()
        else:
            self._run_tick(k)
        self._tick += 1
        return self.engines[(k + 1) % self.slots].gen_ids.t()

    def flush(self, ticks: Optional[Sequence[int]] = None) -> List[Optional[torch.Tensor]]:
        """Drain: finish the batches still in flight with each slot's own per-batch plans; returns their ids oldest
        first (copies), one entry per step of the last `slots - 1` steps. `ticks`: finish only the batches submitted at
        these step indices (None entries for the others) — a server that already answered the older slots skips them."""
        out: List[Optional[torch.Tensor]] = []
        for a in range(min(self.slots - 2, self._tick - 1), -1, -1):    # a = steps since the batch was submitted
            tick = self._tick - 1 - a
            if ticks is not None and tick not in ticks:
                out.append(None)
                continue
            e = self.engines[tick % self.slots]
            j = a - self.lag                                            # decode iterations already done (-1: not prefilled)
            if j < 0:
                ops.run_all(e.prefill_ops)
                j = 0
            for step in e.decode_ops[j:]:
                ops.run_all(step)
            out.append(e.gen_ids.t().clone())
        return out
