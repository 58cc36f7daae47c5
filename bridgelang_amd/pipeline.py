"""Two-deep software pipeline over batches for throughput serving.

A `predict_action` batch has two very different halves: vision + prefill is MFMA-bound (≈ 65 ms at B = 16) but leaves
CUs idle in partial GEMM rounds and between dependent launches; the 6 cached decode steps are HBM-bound weight streaming
(≈ 27 ms). `TwoStagePipeline` keeps two engines (two sets of activations / KV caches over ONE set of weights) and, per
step, runs stage 1 (vision → projector → prefill → first token) of the batch submitted NOW on one stream while stage 2
(decode steps 1..6) of the batch submitted ONE STEP EARLIER runs on a second stream; the step completes one batch.
Work per step is exactly one batch's full computation; only the latency of an individual batch is two steps.
Each of the two stage pairings is captured once as a HIP graph with fork/join stream dependencies.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .engine import OpenVLAEngine
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
