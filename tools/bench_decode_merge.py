"""Decode-iteration cost at full 7B width: 6 x (16 rows through the weight-streaming skinny GEMMs) vs 1 x (96 rows through
the mid-M GEMM) over the real 32 layers + lm_head. Attention excluded (identical in both). python tools/bench_decode_merge.py"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops, weights as W
from bridgelang_amd.ops import EPI_NONE, EPI_RES, EPI_SWIGLU, EPI_F32_BF16R
dev = torch.device("cuda:0")
d = W.openvla_7b_dims()
w = W.allocate(d, dev).fill_synthetic(seed=0)
D, I, V = d.llm_dim, d.llm_inter, d.vocab
z = lambda *s, dtype=torch.bfloat16: torch.zeros(*s, dtype=dtype, device=dev)


ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def plan(M, fused_norm, rows=False):
    """fused_norm: the per-batch skinny path; rows: the merged path in the skinny kernel's arithmetic (bit-identical);
    neither: the round-1 merged path through the tiled mid-M kernels + split-K workspace."""
    G = (lambda *a, **k: ops.gemm(*a, **k)) if fused_norm else (lambda *a, **k: ops.gemm(*a, workspace=ws, **k))
    if rows:
        G = lambda *a, **k: ops.gemm(*a, skinny_rows=True, workspace=ws, **k)
    nrm = ops.rmsnorm_skinny if rows else ops.rmsnorm
    x, h, qkv, ao, act = (torch.randn(M, D, device=dev).to(torch.bfloat16), z(M, D), z(M, 3 * D), z(M, D), z(M, I))
    lg = z(M, V, dtype=torch.float32)
    p = []
    for lw in w.layers:
        if fused_norm:
            p.append(G(x, lw.qkv_w, qkv, EPI_NONE, a_norm=(lw.ln1, d.rms_eps), run=False))
        else:
            p.append(nrm(x, lw.ln1, h, d.rms_eps, run=False))
            p.append(G(h, lw.qkv_w, qkv, EPI_NONE, run=False))
        p.append(G(ao, lw.o_w, x, EPI_RES, res=x, run=False))
        if fused_norm:
            p.append(G(x, lw.gu_w, act, EPI_SWIGLU, a_norm=(lw.ln2, d.rms_eps), run=False))
        else:
            p.append(nrm(x, lw.ln2, h, d.rms_eps, run=False))
            p.append(G(h, lw.gu_w, act, EPI_SWIGLU, run=False))
        p.append(G(act, lw.down_w, x, EPI_RES, res=x, run=False))
    if fused_norm:
        p.append(G(x, w.lm_head, lg, EPI_F32_BF16R, a_norm=(w.norm, d.rms_eps), run=False))
    else:
        p.append(nrm(x, w.norm, h, d.rms_eps, run=False))
        p.append(G(h, w.lm_head, lg, EPI_F32_BF16R, run=False))
    return p


def time_plan(p, reps, inner):
    ops.run_all(p); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            ops.run_all(p)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


wbytes = sum(t.numel() * 2 for lw in w.layers for t in (lw.qkv_w, lw.o_w, lw.gu_w, lw.down_w)) + w.lm_head.numel() * 2
a = time_plan(plan(16, True), 10, 6)
print(f"6 x M=16 skinny : {a:7.2f} ms  ({6 * wbytes / a / 1e6:6.0f} GB/s weight stream)")
for M in (32, 96, 128):
    b = time_plan(plan(M, False), 10, 1)
    c = time_plan(plan(M, False, rows=True), 10, 1)
    print(f"1 x M={M:3d} tiled : {b:7.2f} ms  ({wbytes / b / 1e6:6.0f} GB/s)   skinny-rows: {c:7.2f} ms  ({wbytes / c / 1e6:6.0f} GB/s weight stream)")
# per-shape at M = 96
for name, N, K, epi in (("qkv", 3 * D, D, EPI_NONE), ("o", D, D, EPI_RES), ("gate/up", 2 * I, D, EPI_SWIGLU), ("down", D, I, EPI_RES)):
    M = 96
    xs = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = z(M, N // 2 if epi == EPI_SWIGLU else N)
    attr = {"qkv": "qkv_w", "o": "o_w", "gate/up": "gu_w", "down": "down_w"}[name]
    for tag, kw in (("tiled", dict(workspace=ws)), ("rows ", dict(skinny_rows=True, workspace=ws)), ("rows1", dict(skinny_rows=True))):
        p = [ops.gemm(xs, getattr(lw, attr), out, epi, run=False, **kw, **({"res": out} if epi == EPI_RES else {})) for lw in w.layers]
        t = time_plan(p, 10, 1)
        print(f"  M=96 {tag} {name:8s} N={N:6d} K={K:6d}: {t / 32 * 1e3:7.1f} us  {N * K * 2 / (t / 32) / 1e6:6.0f} GB/s")
xs = torch.randn(96, D, device=dev).to(torch.bfloat16)
hh = z(96, D)
for tag, fn in (("rmsnorm", ops.rmsnorm), ("rmsnorm_skinny", ops.rmsnorm_skinny)):
    t = time_plan([fn(xs, lw.ln1, hh, d.rms_eps, run=False) for lw in w.layers], 10, 1)
    print(f"  M=96 {tag}: {t / 32 * 1e3:6.1f} us")
