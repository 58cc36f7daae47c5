"""A/B of the merged-decode GEMM forms at full 7B width, M = 96 stacked rows: gemm_mid_kernel<SK> (BL_ROWS_STREAM=0) vs
the dispatch with gemm_rows_stream_kernel for the wide layers (o / down stay on the mid kernel either way). Each timing
walks the 32 layers' own weight matrices (cold weights: the stream really comes from HBM) inside one graph.
python tools/bench_rows_stream.py [M]"""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops, weights as W
from bridgelang_amd.ops import EPI_NONE, EPI_RES, EPI_SWIGLU

dev = torch.device("cuda:0")
d = W.openvla_7b_dims()
w = W.allocate(d, dev).fill_synthetic(seed=0)
D, I = d.llm_dim, d.llm_inter
M = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)


def time_plan(p, reps=60):
    ops.run_all(p); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.run_all(p)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


configs = [("mid kernel", "0"), ("rows-stream", "1")] * 2
ref = {}
for name, N, K, epi in (("qkv", 3 * D, D, EPI_NONE), ("o", D, D, EPI_RES), ("gate/up", 2 * I, D, EPI_SWIGLU), ("down", D, I, EPI_RES)):
    xs = torch.randn(M, K, device=dev).to(torch.bfloat16)
    attr = {"qkv": "qkv_w", "o": "o_w", "gate/up": "gu_w", "down": "down_w"}[name]
    n_out = N // 2 if epi == EPI_SWIGLU else N
    res = torch.randn(M, n_out, device=dev).to(torch.bfloat16)
    for tag, on in configs:
        os.environ["BL_ROWS_STREAM"] = on            # read by bl_gemm_skinny_rows_bf16 at every call
        outs = [z(M, n_out) for _ in w.layers]
        p = [ops.gemm(xs, getattr(lw, attr), o, epi, run=False, skinny_rows=True, workspace=ws, **({"res": res} if epi == EPI_RES else {}))
             for lw, o in zip(w.layers, outs)]
        t = time_plan(p)
        got = outs[3].clone()
        ref.setdefault(name, got)
        print(f"M={M} {name:8s} N={N:6d} K={K:6d} {tag:12s}: {t / 32 * 1e3:7.1f} us  {N * K * 2 / (t / 32) / 1e6:6.0f} GB/s  "
              f"bit-identical: {torch.equal(got, ref[name])}", flush=True)
os.environ.pop("BL_ROWS_STREAM", None)
