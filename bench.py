#!/usr/bin/env python3
"""bench.py — action-sequences/s of the OpenVLA-7B hot path on MI355X (BASELINE.json metric, configs[1]).

One "step" = one batch of 16 synthetic 224 px frames + 32-token prompts taken from HBM-resident inputs through the
whole predict_action computation: DINOv2+SigLIP towers → projector → splice → Llama-2-7B prefill (S = 288) → 6 cached
decode steps with on-device greedy argmax → 7 action-token ids per sequence. Nothing is skipped or cached between
steps; weights are seeded-synthetic bf16 (no checkpoint exists offline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (N > 1: one replica per GPU)
    python bench.py --mode train --stage vla-full-train|vla-train|lora [--gpus N]     (BASELINE configs[2] / [3])

`--gpus N` with N > 1 and no torchrun environment: this process — before it touches the GPU — starts N children (one per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) and relays rank 0's JSON line; under torchrun
the ranks already exist and `--gpus` must equal WORLD_SIZE.

Inference shards by sequence with no exchange step ("replicas only"): every rank runs its own batch, the
timed region is bracketed by barrier + synchronize, the elapsed time is the MAX over ranks. Training shards the batch
by rank with the optimizer state partitioned (training/sharding.py: reduce-scatter of gradient buckets overlapped with
the backward, all-gather of updated weights) — RCCL over xGMI.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = the tiled MFMA GEMM, measured live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded, stated sample).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent


def newest_pmc(name: str):
    """The most recent committed PMC summary profiles/pmc_rNN*/<name> (bench.py cannot run rocprofv3 on itself: the traffic
    figure it quotes is measured separately, and `traffic_source` says when and on which commit)."""
    cands = sorted((ROOT / "profiles").glob(f"pmc_r*/{name}"), key=lambda q: q.parent.name)
    return cands[-1] if cands else None


def pmc_provenance(path, pj: dict) -> dict:
    c = pj.get("collected") or {}
    return {"file": str(path.relative_to(ROOT)), "collected_at_commit": c.get("commit"), "collected_on": c.get("date"),
            "round": c.get("round"), "note": "valid while the tile GEMM kernels (gemm256s / gemm288s / tail) are as at that commit; otherwise re-run the --pmc passes (tools/pmc_gemm.py)"}
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
    ap.add_argument("--batch", type=int, default=None, help="per GPU; default 16 (infer, lora) / 32 (full fine-tune, conf/vla.py:83)")
    ap.add_argument("--prompt-len", type=int, default=32)
    ap.add_argument("--model", default="openvla-7b", choices=["openvla-7b", "openvla-tiny", "prism-13b"])
    ap.add_argument("--no-graph", action="store_true", help="replay the op plan eagerly instead of as one HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="Llama prefill projections as W8A8 e4m3 GEMMs (BASELINE configs[4] extension; not the bf16 headline)")
    ap.add_argument("--fp8-wgrad", action="store_true", help="--mode train --fp8: the decoder layers' weight-gradient GEMMs on the e4m3 path too")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="train: one optimisation step of run_vla_training per bench step (BASELINE configs[2] / [3])")
    ap.add_argument("--stage", default="vla-full-train", choices=["vla-full-train", "vla-train", "lora"],
                    help="--mode train: full fine-tune (configs[2]), vision frozen, or LoRA r=32 (configs[3])")
    ap.add_argument("--reduce", default="fp32", choices=["fp32", "bf16"], help="--mode train: gradient wire format")
    ap.add_argument("--recompute", action="store_true",
                    help="--mode train: keep only the decoder layers' inputs and replay each layer in backward (fsdp.py:171-183)")
    ap.add_argument("--shard-params", action="store_true",
                    help="--mode train: FSDP FULL_SHARD for the decoder layers (parameters sharded over the ranks, gathered per layer)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / fence / JSON plumbing only (gloo, no GPU, a sleep per step): the CPU test of --gpus N")
    ap.add_argument("--train-legs", default="auto", choices=["auto", "on", "off"],
                    help="--mode infer: after the inference measurement (engines freed) also time 3 steps each of BASELINE "
                         "configs[2]'s per-GPU shape (vla-full-train, B = 32) and configs[3] (LoRA r = 32, B = 16) and report them "
                         "under end_to_end.train_cfg2_n1 / train_cfg3_lora_n1 of the same JSON line; auto = the default 1-GPU "
                         "openvla-7b run only")
    ap.add_argument("--pipeline", type=int, default=7, choices=[1, 2, 7, 8],
                    help="2 = overlap batch i's decode with batch i+1's vision+prefill (TwoStagePipeline); 7 = StaggeredDecodePipeline "
                         "(one merged decode iteration over the 6 older batches per step)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 32 if (args.mode == "train" and args.stage != "lora") else 16
    return args


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """Parent of a `--gpus N` run without torchrun: N children of this same command line, one per GPU. The parent makes
    no GPU call (a process that has initialised the GPU must not be replaced or forked on this pool); it relays rank 0's
    stdout (the JSON line) and every rank's stderr, and exits with the worst child status."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, str(Path(__file__).resolve())] + sys.argv[1:]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL) for r in range(n)]
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out.decode().splitlines():          # stdout carries exactly the JSON line; library chatter goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


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


def cpu_baseline_whole(dims, ids0, pv0, gpu_ids0) -> dict:
    """The WHOLE CPU oracle (oracle/restate.py: both towers, projector, 32-layer prefill, 6 cached decode steps, greedy) for
    ONE sequence of the workload — sequence 0 of rank 0's batch, on the same synthetic checkpoint the GPU ran — timed once
    on this host's cores (used when the host has >= 64 cores and 64 GB free: the checkpoint is held as fp32, 30 GB;
    otherwise the per-block extrapolation below). Also reports whether its 7 ids equal the GPU's."""
    from bridgelang_amd import weights as W
    from oracle import pick_threads, restate as R, synth as S
    pick_threads()
    torch.set_flush_denormal(True)
    t0 = time.perf_counter()
    sd = S.synth_state_dict(W.tensor_specs(dims, "init"), seed=0)
    for k in list(sd):                      # "model load": keep the (bf16-valued) weights as fp32 tensors, so the timed region
        sd[k] = sd[k].float()               # is arithmetic, not 7 bf16→fp32 conversions of the checkpoint (30 GB of host RAM)
    t_ckpt = time.perf_counter() - t0
    model = R.OracleModel.from_dims(sd, dims)
    t0 = time.perf_counter()
    with torch.no_grad():
        gen, _ = model.generate(ids0, pv0, n_new=7, last_row_only=True)
    per_seq = time.perf_counter() - t0
    same = gen[0].tolist() == list(gpu_ids0)
    return {"value": 1.0 / per_seq, "unit": "action-seqs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": (f"oracle/restate.py, the whole {dims.name} path for ONE sequence (sequence 0 of the batch: both towers, "
                       f"projector, Llama prefill S={ids0.shape[1] + dims.n_patches}, 6 cached decode steps, greedy) timed once: "
                       f"{per_seq:.1f} s (+ {t_ckpt:.0f} s to generate the synthetic checkpoint on the host and hold it as fp32, not timed); "
                       f"oracle ids {gen[0].tolist()} {'==' if same else '!='} GPU ids (random-init logits are nearly flat: a "
                       f"difference is a near tie, tests/test_cfg_7b_golden_gpu.py)"),
            "ids_equal_gpu": same}


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


def dry_run(args, rank: int, world: int) -> None:
    """The launch contract without a GPU: rendezvous (gloo), warm-up, EXACTLY K timed 'steps' (a 10 ms sleep each)
    between fences, max over ranks, one JSON line from rank 0."""
    from bridgelang_amd import replicas
    distributed = replicas.init("gloo")
    for _ in range(args.warmup):
        time.sleep(0.01)
    replicas.fence(None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))            # the slowest rank sets the time
    replicas.fence(None)
    elapsed = replicas.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "dry-run steps/s", "value": round(world * args.batch * args.steps / elapsed, 3), "unit": "items/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none", "config": {"workload": "dry-run", "mode": args.mode}}),
              flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def train_main(args, rank: int, world: int, dev, distributed: bool) -> None:
    line = train_line(args, rank, world, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def train_line(args, rank: int, world: int, dev) -> dict:
    """BASELINE configs[2] (full fine-tune, per-GPU batch 32, S = 296, sharded optimizer over the ranks) / configs[3]
    (LoRA r = 32, batch 16): one `run_vla_training` iteration per step — forward, backward (bucket reduce-scatter
    overlapped), global-norm clip, AdamW, weight re-pack / all-gather (base_strategy.py:284-366, fsdp.py:135-270).
    Returns the JSON record (rank 0; None elsewhere)."""
    from bridgelang_amd import replicas, weights as W
    from bridgelang_amd.training.step import TrainStep
    dims = {"openvla-7b": W.openvla_7b_dims, "prism-13b": W.prism_13b_dims, "openvla-tiny": W.tiny_dims}[args.model]()
    w = W.allocate(dims, dev).fill_synthetic(seed=0)
    lora = None
    if args.stage == "lora":
        from bridgelang_amd.training.lora import LoraAdapters
        lora = LoraAdapters(w, r=32)
    B, L = args.batch, args.prompt_len + 8                # 32 prompt ids + 7 action ids + EOS (SURVEY §8d cfg 3 / 4)
    ts = TrainStep(w, args.stage, B, L, max_grad_norm=1.0 if lora is None else float("inf"),
                   weight_decay=0.0 if lora is None else 0.01, lora=lora, world=world, rank=rank,
                   reduce_dtype=torch.float32 if args.reduce == "fp32" else torch.bfloat16, recompute=args.recompute,
                   shard_params=args.shard_params, fp8=args.fp8, fp8_wgrad=getattr(args, "fp8_wgrad", False))
    g = torch.Generator().manual_seed(100 + rank)         # each rank draws its own stream (base_strategy.py:259-266)
    ids = torch.randint(3, 31000, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
    ids[:, -1] = 2
    labels = torch.full((B, L), -100)
    labels[:, -8:] = ids[:, -8:]
    img = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).float().div_(255.0).permute(0, 3, 1, 2)
    m = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    sd = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    pv = torch.cat([(img - m) / sd, (img - 0.5) / 0.5], dim=1).to(torch.bfloat16)
    ts.set_batch(ids, None, pv, labels)                   # batch resident in HBM before the timed region
    lr = 2e-5 if lora is None else 5e-4
    graph = not args.no_graph      # at world > 1 TrainStep keeps the backward (per-bucket collectives) and the parameter-sharded
                                   # plans eager by itself; forward, vision forward and the re-pack plan still replay as HIP graphs
    losses = []
    for _ in range(max(args.warmup, 1)):
        loss, _ = ts.step(lr, graph=graph)
        losses.append(float(loss.item()))
    replicas.fence(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = ts.step(lr, graph=graph)
    replicas.fence(dev)
    elapsed = replicas.max_over_ranks(time.perf_counter() - t0, dev)
    losses.append(float(loss.item()))
    if rank == 0:
        # per-phase and GEMM-family timing: one instrumented eager step outside the timed region (HIP events on the
        # launch stream); roofline = the tiled MFMA GEMM family (forward, dgrad, wgrad) per bl_gemm_bf16 call
        ev = lambda: torch.cuda.Event(enable_timing=True)
        e = [ev() for _ in range(4)]
        torch.cuda.synchronize()
        e[0].record(); ts.forward(False)
        e[1].record(); ts.backward(False)
        e[2].record(); ts.clip_grad_norm(); ts.optimizer_step(lr, False)
        e[3].record()
        torch.cuda.synchronize()
        phases = {k: round(e[i].elapsed_time(e[i + 1]), 3) for i, k in enumerate(("forward", "backward", "clip+adamw+repack"))}
        plan = list(ts.vision_forward_ops) + list(ts.forward_ops) + list(ts.backward_ops)
        if not ts.train_vision:
            plan = list(ts._vis.vision_ops) + plan
        stream = torch.cuda.current_stream()
        evs = [ev() for _ in range(len(plan) + 1)]
        sptr = stream.cuda_stream
        evs[0].record(stream)
        for i, op in enumerate(plan):
            assert op.fn(*op.args, sptr) == 0, op.name
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        agg: dict = {}
        for i, op in enumerate(plan):
            a = agg.setdefault(op.name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["launches"] += 1; a["ms"] += evs[i].elapsed_time(evs[i + 1]); a["flops"] += op.flops; a["bytes"] += op.bytes
        # the tiled MFMA GEMM family: NT (forward, dgrad), TN (wgrad) and, with --fp8, the e4m3 forward / dgrad launches
        fam = [agg[k] for k in ("bl_gemm_bf16", "bl_gemm_tn_bf16", "bl_gemm_fp8") if k in agg]
        gemm = {f: sum(a[f] for a in fam) for f in ("ms", "flops", "launches")}
        achieved = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
        model_flops = sum(op.flops for op in plan)
        ms = elapsed / args.steps * 1e3
        train_traffic = None          # HBM bytes per forward GEMM call from the committed PMC passes (same M as this run only)
        pmc = newest_pmc("gemm_traffic_train.json")
        pmc_src = None
        if pmc is not None and args.model == "openvla-7b" and ts.T == 9472 and not args.fp8:
            pj = json.loads(pmc.read_text())
            train_traffic, pmc_src = round(pj["avg_hbm_bytes_per_call_llama_layer"]), pmc_provenance(pmc, pj)
        cfg_no = 4 if args.model == "prism-13b" else 3 if args.stage == "lora" else 2
        line = {
            "metric": f"samples/sec {dims.name} {args.stage} bf16", "value": round(world * B * args.steps / elapsed, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"{dims.name} {args.stage} (BASELINE configs[{cfg_no}]): per-GPU batch {B} x S={ts.S} "
                                    f"DummyDataset-shaped samples (32 prompt ids + 7 action ids + EOS, 224px frame), forward + "
                                    f"backward + global-norm clip + AdamW" + (" on LoRA r=32 adapters" if lora is not None else "")
                                    + (f"; optimizer state sharded over {world} ranks, {args.reduce} gradient reduce-scatter "
                                       f"per bucket overlapped with backward, bf16 weight all-gather" if world > 1 else "")),
                       "batch_per_gpu": B, "global_batch": B * world, "seq_len": ts.S, "stage": args.stage,
                       "parallelism": (f"dp{world} sharded-optimizer ({'full-shard: parameters and gradients of all FSDP units sharded' if args.shard_params else 'shard-grad-op'})"
                                       if world > 1 else "single GPU"),
                       "hip_graph": graph, "recompute_activations": bool(args.recompute), "shard_params": bool(args.shard_params), "fp8_fwd_dgrad": bool(args.fp8),
                       "fp8_wgrad": bool(getattr(args, "fp8_wgrad", False)),
                       "fp8_wgrad_gemms": {"e4m3": ts.fp8_wgrad_gemms[0], "bf16_fallback": ts.fp8_wgrad_gemms[1]}},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": train_traffic,
                         "traffic_note": "avg HBM+Infinity-Cache bytes per forward bl_gemm_bf16 call over the 4 decoder-layer GEMM shapes at "
                                         "M = 9472 token rows, separate --pmc passes (committed summary, see traffic_source); dgrad / wgrad "
                                         "launches of the family not counted" if train_traffic else None,
                         "traffic_source": pmc_src,
                         "kernel": "tiled MFMA GEMM family per call: bl_gemm_bf16 (forward, dgrad), bl_gemm_tn_bf16 (wgrad)" + (", bl_gemm_fp8 (e4m3 forward / dgrad; priced against the bf16 peak)" if args.fp8 else ""),
                         "launches_per_step": gemm["launches"], "avg_launch_us": round(gemm["ms"] * 1e3 / gemm["launches"], 2),
                         "algorithmic_gflop_per_launch": round(gemm["flops"] / gemm["launches"] / 1e9, 3)},
            "end_to_end": {"model_tflop_per_step_per_gpu": round(model_flops / 1e12, 2),
                           "mfma_util_whole_step": round(model_flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                           "phases_ms_eager_events": phases,
                           "per_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])},
                           "hbm_bound_kernels_GBs": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9) for k, v in
                                                     sorted(agg.items(), key=lambda kv: -kv[1]["ms"]) if v["bytes"] and not v["flops"]},
                           "trainable_params_B": round(ts.store.n_params / 1e9, 3),
                           "hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                           "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)]},
        }
        return line
    return None


def train_legs(args, dev) -> dict:
    """BASELINE configs[2] / configs[3] on the same clock as the headline: after the inference engines are freed, 1 warm-up
    + 3 timed steps each of the full fine-tune at its per-GPU shape (B = 32, S = 296; base_strategy.py:284-366) and of the
    LoRA r = 32 step (B = 16; vla-scripts/finetune.py:253-318), through the same train_line() `--mode train` prints. A leg
    that fails (e.g. out of memory on a box with less free HBM) is reported as {"error": …}; it never costs the headline."""
    import copy
    import gc
    out = {}
    for key, stage, batch in (("train_cfg2_n1", "vla-full-train", 32), ("train_cfg3_lora_n1", "lora", 16)):
        a = copy.copy(args)
        a.mode, a.stage, a.batch, a.steps, a.warmup = "train", stage, batch, 3, 1
        a.recompute = a.shard_params = a.fp8 = a.fp8_wgrad = False
        a.reduce = "fp32"
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        try:
            ln = train_line(a, 0, 1, dev)
            e2e = ln["end_to_end"]
            out[key] = {"ms_per_step": ln["ms_per_step"], "samples_per_s": ln["value"], "steps": 3, "warmup": 1,
                        "mfma_util_whole_step": e2e["mfma_util_whole_step"], "model_tflop_per_step": e2e["model_tflop_per_step_per_gpu"],
                        "gemm_family_tflops_per_call": ln["roofline"]["achieved"], "phases_ms_eager_events": e2e["phases_ms_eager_events"],
                        "hbm_gib": e2e["hbm_gib"], "loss_first_last": e2e["loss_first_last"], "workload": ln["config"]["workload"],
                        "leg_wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as exc:                                      # noqa: BLE001 — report, keep the headline
            out[key] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}", "leg_wall_s": round(time.perf_counter() - t0, 1)}
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))          # parent: no GPU call before or after this line
    from bridgelang_amd import replicas
    rank, local, world = replicas.env_rank()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = replicas.init("nccl", dev)      # RCCL; inference: only the timing fence uses it (replicas, no data-path collective)
    if args.mode == "train":
        return train_main(args, rank, world, dev, distributed)

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
    ids_cpu0, pv_cpu0 = ids[:1].cpu(), pv[:1].cpu()
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
        pmc = newest_pmc("gemm_traffic.json")
        pmc_src = None
        if pmc is not None and args.model == "openvla-7b" and args.batch == 16 and args.prompt_len == 32 and not args.fp8:
            pj = json.loads(pmc.read_text())
            traffic, pmc_src = round(pj["avg_hbm_bytes_per_call_llama_layer"]), pmc_provenance(pmc, pj)
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
                         "traffic_note": "avg HBM+Infinity-Cache bytes per bl_gemm_bf16 call over the 4 Llama prefill GEMMs, separate --pmc passes: "
                                         "read from a COMMITTED summary (bench.py cannot run rocprofv3 on itself), see traffic_source",
                         "traffic_source": pmc_src,
                         "kernel": ("gemm256s_fp8_kernel (per bl_gemm_fp8 call)" if args.fp8 else
                                    "tiled MFMA GEMM family: gemm256s_kernel / gemm288s_kernel + gemm_tail_kernel (per bl_gemm_bf16 call)"), "launches_per_step": gemm["launches"],
                         "avg_launch_us": round(gemm["ms"] * 1e3 / gemm["launches"], 2),
                         "algorithmic_gflop_per_launch": round(gemm["flops"] / gemm["launches"] / 1e9, 3),
                         "sustained_mfma_note": ("`peak` is the dense bf16 spec figure; a register-only stream of the same MFMA on "
                                                 "N(0,1) operands sustains 1611 TFLOP/s on this pool (power cap; tools/micro/"
                                                 "w4_ceiling.hip, DESIGN §3)" if not args.fp8 else None)},
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
        legs = args.train_legs == "on" or (args.train_legs == "auto" and world == 1 and args.model == "openvla-7b"
                                           and not args.fp8 and args.pipeline == 7 and args.batch == 16)
        if legs and world == 1:          # training on the same clock: free the inference state first
            if args.pipeline != 1:
                del pipe
            del eng, w, prof
            line["end_to_end"].update(train_legs(args, dev))
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would wait at the barrier)
            big_host = (os.cpu_count() or 1) >= 64 and os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") > (64 << 30)
            if big_host and args.model == "openvla-7b":
                line["cpu_baseline"] = cpu_baseline_whole(dims, ids_cpu0, pv_cpu0, ids_out[0].tolist())
            else:
                line["cpu_baseline"] = cpu_baseline(dims, args.batch, args.prompt_len)
        print(json.dumps(line), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
