#!/usr/bin/env python3
"""bench.py — action-sequences/s of the OpenVLA-7B hot path on MI355X (BASELINE.json metric, configs[1]).

One "step" = one batch of 16 synthetic 224 px frames + 32-token prompts taken from HBM-resident inputs through the
whole predict_action computation: DINOv2+SigLIP towers → projector → splice → Llama-2-7B prefill (S = 288) → 6 cached
decode steps with on-device greedy argmax → 7 action-token ids per sequence. Nothing is skipped or cached between
steps; weights are seeded-synthetic bf16 (no checkpoint exists offline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (N > 1: one replica per GPU)

Inference shards by sequence with no exchange step ("replicas only"): every rank runs its own batch, the
timed region is bracketed by barrier + synchronize, the elapsed time is the MAX over ranks.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = the tiled MFMA GEMM, measured live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded, stated sample).
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (guides/MI355X_MICROARCH.md)
FP8_MFMA_PEAK_TFLOPS = 5000.0      # dense fp8 through the block-scaled MFMA (same table)
HBM_PEAK_GBS = 8000.0
ALGO_TFLOP_PER_SEQ = 4.238         # SURVEY §8d / BASELINE.md §2, openvla-7b, S = 288, KV cache, last-row lm_head


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--prompt-len", type=int, default=32)
    ap.add_argument("--model", default="openvla-7b", choices=["openvla-7b", "openvla-tiny", "prism-13b"])
    ap.add_argument("--no-graph", action="store_true", help="replay the op plan eagerly instead of as one HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="Llama prefill projections as W8A8 e4m3 GEMMs (BASELINE configs[4] extension; not the bf16 headline)")
    ap.add_argument("--pipeline", type=int, default=7, choices=[1, 2, 7, 8],
                    help="2 = overlap batch i's decode with batch i+1's vision+prefill (TwoStagePipeline); 7 = StaggeredDecodePipeline "
                         "(one merged decode iteration over the 6 older batches per step)")
    return ap.parse_args()


def make_inputs(B: int, L: int, seed: int, device):
    """SURVEY §8d cfg 2: uniform uint8 frames → DINOv2 (ImageNet) and SigLIP (0.5/0.5) normalisation → [B,6,224,224]
    bf16; prompts = BOS + 30 ids uniform in [3, 31743] + 29871."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).float().div_(255.0).permute(0, 3, 1, 2)
    m = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    s = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    pv = torch.cat([(img - m) / s, (img - 0.5) / 0.5], dim=1).to(torch.bfloat16)
    ids = torch.randint(3, 31743, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 1, 29871
    return ids.to(device), pv.to(device)


def per_kernel_profile(eng) -> dict:
    """One instrumented eager pass: HIP events (on the launch stream = torch's current stream) around every launch,
    aggregated per C-ABI entry point. Outside the timed region."""
    plan = eng.all_ops()
    stream = torch.cuda.current_stream()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(plan) + 1)]
    s = stream.cuda_stream
    torch.cuda.synchronize()
    evs[0].record(stream)
    for i, op in enumerate(plan):
        rc = op.fn(*op.args, s)
        assert rc == 0, op.name
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    agg: dict = {}
    for i, op in enumerate(plan):
        a = agg.setdefault(op.name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        a["launches"] += 1
        a["ms"] += evs[i].elapsed_time(evs[i + 1])
        a["flops"] += op.flops
        a["bytes"] += op.bytes
    return agg


def cpu_baseline(dims, B: int, L: int) -> dict:
    """The CPU oracle (oracle/restate.py, fp32 arithmetic on bf16-rounded values, KV cache, greedy) timed on this
    host's cores on a bounded sample of the same workload: ONE block of each ViT tower, the projector, ONE Llama layer
    (prefill at S = 288 and one cached decode step) and the last-row lm_head, all at full 7B width for ONE sequence;
    per-sequence time = Σ component × its repeat count in the full model. Random weights (timing does not depend on
    values); extrapolation is by exact layer counts, stated in `sample`."""
    from oracle import pick_threads, restate as R
    pick_threads()                       # all cores on a real host; fewer where OpenMP barriers dominate (VMs)
    torch.set_flush_denormal(True)       # synthetic weights give |score| ~ 70: exp underflow → denormal stalls on x86
    p = R.Prec(True)
    S = L + dims.n_patches
    g = torch.Generator().manual_seed(0)
    rnd = lambda *shape: (torch.randn(*shape, generator=g) * 0.02)

    def timed(fn, reps=2):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    def vit_block_time(t):
        D, H, T = t.dim, t.mlp, t.tokens
        w = dict(n1w=torch.ones(D), n1b=torch.zeros(D), qkv=rnd(3 * D, D), qb=rnd(3 * D), pw=rnd(D, D), pb=rnd(D),
                 f1=rnd(H, D), f1b=rnd(H), f2=rnd(D, H), f2b=rnd(D))
        x = rnd(1, T, D) * 50
        hd = t.head_dim

        def run():
            h = R.layernorm(p, x, w["n1w"], w["n1b"], 1e-6)
            qkv = R.linear(p, h, w["qkv"], w["qb"]).view(1, T, 3, t.heads, hd).permute(2, 0, 3, 1, 4)
            a = R.attention(p, qkv[0], qkv[1], qkv[2], hd ** -0.5, False).permute(0, 2, 1, 3).reshape(1, T, D)
            y = p.rb(x + R.linear(p, a, w["pw"], w["pb"]))
            h = R.layernorm(p, y, w["n1w"], w["n1b"], 1e-6)
            f = R.gelu(p, R.linear(p, h, w["f1"], w["f1b"]))
            return p.rb(y + R.linear(p, f, w["f2"], w["f2b"]))
        return timed(run)

    t_dino, t_sig = vit_block_time(dims.dino), vit_block_time(dims.siglip)
    V, D, I = dims.vision_dim, dims.llm_dim, dims.llm_inter
    pw = [rnd(4 * V, V), rnd(D, 4 * V), rnd(D, D)]
    feats = rnd(1, 256, V) * 50
    t_proj = timed(lambda: R.linear(p, R.gelu(p, R.linear(p, R.gelu(p, R.linear(p, feats, pw[0])), pw[1])), pw[2]))
    lm = "language_model.model"
    sd = {f"{lm}.layers.0.input_layernorm.weight": torch.ones(D), f"{lm}.layers.0.post_attention_layernorm.weight": torch.ones(D),
          f"{lm}.norm.weight": torch.ones(D), "language_model.lm_head.weight": rnd(dims.vocab, D)}
    for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
        sd[f"{lm}.layers.0.self_attn.{n}.weight"] = rnd(D, D)
    sd[f"{lm}.layers.0.mlp.gate_proj.weight"], sd[f"{lm}.layers.0.mlp.up_proj.weight"] = rnd(I, D), rnd(I, D)
    sd[f"{lm}.layers.0.mlp.down_proj.weight"] = rnd(D, I)
    x = rnd(1, S, D) * 50
    last = torch.tensor([S - 1])
    cache_box = {}

    def prefill():
        _, cache_box["c"] = R.llama_forward(p, sd, x, dims.llm_heads, 1, dims.rms_eps, dims.rope_theta, rows=last)
    t_pre_plus_head = timed(prefill)
    xd = rnd(1, 1, D) * 50
    t_dec_plus_head = timed(lambda: R.llama_forward(p, sd, xd, dims.llm_heads, 1, dims.rms_eps, dims.rope_theta,
                                                    cache=cache_box["c"]))
    h1 = rnd(1, 1, D)
    t_head = timed(lambda: R.linear(p, R.rmsnorm(p, h1, sd[f"{lm}.norm.weight"], 1e-6), sd["language_model.lm_head.weight"]))
    t_pre, t_dec = max(t_pre_plus_head - t_head, 0.0), max(t_dec_plus_head - t_head, 0.0)
    n_dec = 6
    per_seq = (dims.dino.n_run * t_dino + dims.siglip.n_run * t_sig + t_proj + dims.llm_layers * t_pre
               + n_dec * dims.llm_layers * t_dec + (n_dec + 1) * t_head)
    return {"value": 1.0 / per_seq, "unit": "action-seqs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": (f"oracle/restate.py at full {dims.name} width, batch 1, S={S}: 1 DINOv2 block ({t_dino * 1e3:.0f} ms) "
                       f"x{dims.dino.n_run}, 1 SigLIP block ({t_sig * 1e3:.0f} ms) x{dims.siglip.n_run}, projector "
                       f"({t_proj * 1e3:.0f} ms), 1 Llama layer prefill ({t_pre * 1e3:.0f} ms) x{dims.llm_layers}, 1 layer "
                       f"cached decode ({t_dec * 1e3:.0f} ms) x{n_dec * dims.llm_layers}, lm_head row ({t_head * 1e3:.0f} ms) "
                       f"x{n_dec + 1}; extrapolated {per_seq:.1f} s per sequence")}


def main() -> None:
    args = parse()
    from bridgelang_amd import replicas
    rank, local, world = replicas.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = replicas.init("nccl", dev)      # RCCL; only the timing fence uses it (replicas, no data-path collective)

    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    dims = {"openvla-7b": W.openvla_7b_dims, "prism-13b": W.prism_13b_dims, "openvla-tiny": W.tiny_dims}[args.model]()
    w = W.allocate(dims, dev).fill_synthetic(seed=0)
    ids, pv = make_inputs(args.batch, args.prompt_len, seed=rank, device=dev)

    def timed(replay, steps):
        """EXACTLY `steps` steps bracketed by barrier + synchronize; elapsed = MAX over ranks."""
        replicas.fence(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            replay()
        replicas.fence(dev)
        return replicas.max_over_ranks(time.perf_counter() - t0, dev)

    # ---- one batch in flight (the latency-oriented predict_action path) ----
    eng = OpenVLAEngine(w, args.batch, args.prompt_len, fp8=args.fp8)
    eng.set_inputs(ids, pv)                      # inputs resident in HBM before the timed region
    if args.no_graph:
        eng.run_eager()
    else:
        eng.capture()
    for _ in range(args.warmup):
        eng.replay()
    elapsed_single = timed(eng.replay, args.steps)
    elapsed = elapsed_single
    # ---- two batches in flight: batch i's decode overlaps batch i+1's vision + prefill (throughput serving) ----
    if args.pipeline == 2:
        from bridgelang_amd.pipeline import TwoStagePipeline
        pipe = TwoStagePipeline(w, args.batch, args.prompt_len)
        for e in pipe.engines:
            e.set_inputs(ids, pv)
        if not args.no_graph:
            pipe.capture()
        for _ in range(max(args.warmup, 2)):     # also fills the pipeline
            pipe.step()
        elapsed = timed(pipe.step, args.steps)   # every step completes one batch (submitted one step earlier)
    elif args.pipeline >= 7:
        from bridgelang_amd.pipeline import StaggeredDecodePipeline
        pipe = StaggeredDecodePipeline(w, args.batch, args.prompt_len, split_vision=args.pipeline == 8, fp8=args.fp8)
        for e in pipe.engines:
            e.set_inputs(ids, pv)
        if not args.no_graph:
            pipe.capture()
        for _ in range(max(args.warmup, pipe.slots)):     # also fills the pipeline
            pipe.step()
        elapsed = timed(pipe.step, args.steps)   # every step submits one batch and completes the one submitted 6 steps earlier
        ids_pipe = pipe.step().clone().cpu()
    ids_out = eng.gen_ids.t().cpu()
    if args.pipeline >= 7 and rank == 0:
        pipe_agree = float((ids_pipe == ids_out).all(dim=1).float().mean())

    if rank == 0:
        seqs = world * args.batch * args.steps
        value = seqs / elapsed
        prof = per_kernel_profile(eng)
        gemm = prof["bl_gemm_fp8" if args.fp8 else "bl_gemm_bf16"]     # the dominant kernel of the mode measured
        peak = FP8_MFMA_PEAK_TFLOPS if args.fp8 else BF16_MFMA_PEAK_TFLOPS
        achieved = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
        skinny = prof.get("bl_gemm_skinny_bf16")
        kern_ms = sum(a["ms"] for a in prof.values())
        traffic = None      # HBM bytes per GEMM call from the committed PMC passes (bench.py cannot run rocprofv3 itself)
        pmc = ROOT / "profiles" / "pmc_r01" / "gemm_traffic.json"
        if pmc.exists() and args.model == "openvla-7b" and args.batch == 16 and args.prompt_len == 32 and not args.fp8:
            traffic = round(json.loads(pmc.read_text())["avg_hbm_bytes_per_call_llama_layer"])
        # algorithmic work per sequence: the SURVEY figure for the BASELINE model, the plan's own GEMM + attention FLOPs otherwise
        algo = ALGO_TFLOP_PER_SEQ if args.model == "openvla-7b" else sum(op.flops for op in eng.all_ops()) / args.batch / 1e12
        line = {
            "metric": f"action-seqs/sec (7-DoF, 224px) {dims.name} {'fp8' if args.fp8 else 'bf16'}", "value": round(value, 3), "unit": "action-seqs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8-e4m3 (W8A8 Llama prefill projections) + bf16" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": (f"{dims.name} bf16 inference (BASELINE configs[1]): batch {args.batch} synthetic 224px "
                                    f"frames + {args.prompt_len}-token prompts per GPU, predict_action = vision towers + "
                                    f"projector + Llama prefill S={eng.S} + 6 cached decode steps, greedy"
                                    + {1: "", 2: "; 2 batches in flight (batch i decode overlaps batch i+1 vision+prefill), one "
                                                 "batch completes per step",
                                       7: "; 7 batches in flight: per step one batch is submitted (vision+prefill) while decode "
                                          "iteration g of the batch submitted g steps earlier, g = 1..6, runs as ONE merged 96-row "
                                          "pass over the weights (bit-identical to per-batch decode); one batch completes per step"}.get(args.pipeline, "8 in flight")),
                       "batch_per_gpu": args.batch, "prompt_len": args.prompt_len, "seq_len": eng.S,
                       "replicas": world, "hip_graph": not args.no_graph, "pipeline_depth": args.pipeline},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_note": "avg HBM+Infinity-Cache bytes per bl_gemm_bf16 call over the 4 Llama prefill GEMMs, separate --pmc passes (profiles/pmc_r01/gemm_traffic.json)",
                         "kernel": ("gemm256s_fp8_kernel (per bl_gemm_fp8 call)" if args.fp8 else
                                    "gemm256s_kernel + gemm_tail_kernel / gemm128_kernel (per bl_gemm_bf16 call)"), "launches_per_step": gemm["launches"],
                         "avg_launch_us": round(gemm["ms"] * 1e3 / gemm["launches"], 2),
                         "algorithmic_gflop_per_launch": round(gemm["flops"] / gemm["launches"] / 1e9, 3)},
            "end_to_end": {"algorithmic_tflop_per_seq": round(algo, 3),
                           "mfma_util_whole_step": round(value / world * algo / BF16_MFMA_PEAK_TFLOPS, 4),
                           "kernel_ms_per_step_eager_events": round(kern_ms, 3),
                           "per_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                           "decode_weight_stream_GBs": (round(skinny["bytes"] / (skinny["ms"] * 1e-3) / 1e9, 1) if skinny else None),
                           "decode_hbm_frac": (round(skinny["bytes"] / (skinny["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if skinny else None),
                           "one_batch_in_flight": {"value": round(seqs / elapsed_single, 3),
                                                   "ms_per_step": round(elapsed_single / args.steps * 1e3, 3)},
                           "first_ids": ids_out[0].tolist()},
        }
        if args.pipeline >= 7:   # same inputs in every slot: fraction of sequences whose 7 ids equal the one-batch engine's
            line["end_to_end"]["staggered_ids_equal_one_batch_engine"] = round(pipe_agree, 4)
            line["end_to_end"]["staggered_ids_note"] = ("the merged 96-row decode pass runs bl_gemm_skinny_rows_bf16 + bl_rmsnorm_skinny_bf16, "
                                                       "which reproduce the per-batch weight-streaming kernels' fp32 summation order: ids "
                                                       "and logits are bit-identical to the one-batch engine "
                                                       "(tests/test_full_size_gpu.py::test_staggered_pipeline_full_size)")
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would wait at the barrier)
            line["cpu_baseline"] = cpu_baseline(dims, args.batch, args.prompt_len)
        print(json.dumps(line), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
