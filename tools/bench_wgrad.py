#!/usr/bin/env python3
"""Weight-gradient GEMM: the TN kernel (dy, x read in place) against the round-1 form (two transposes + NT GEMM) on the
Llama-2-7B layer shapes at the training batch (T = 32 x 296 tokens).  python tools/bench_wgrad.py [T]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bridgelang_amd import ops, train_ops as T
from bridgelang_amd.ops import EPI_F32

dev = torch.device("cuda:0")
Tn = int(sys.argv[1]) if len(sys.argv) > 1 else 9472
Tp = (Tn + 63) // 64 * 64
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot_tn = tot_nt = 0.0
for name, N, K in shapes:
    dy = torch.randn(Tn, N, device=dev).to(torch.bfloat16)
    x = torch.randn(Tn, K, device=dev).to(torch.bfloat16)
    out = torch.empty(N, K, dtype=torch.float32, device=dev)
    tA = torch.empty(N, Tp, dtype=torch.bfloat16, device=dev)
    tBp = torch.empty(K // 16, Tp // 32, 64, 8, dtype=torch.bfloat16, device=dev)
    tn = T.gemm_tn(dy, x, out, workspace=ws, run=False)
    nt = [T.transpose_pad(dy, tA, Tp, run=False), T.transpose_pack(x, tBp, Tp, run=False),
          ops.gemm(tA, tBp, out, EPI_F32, workspace=ws, run=False)]
    us_tn = timeit(tn.run)
    us_nt = timeit(lambda: ops.run_all(nt))
    us_g = timeit(nt[2].run)
    fl = 2.0 * Tn * N * K
    tot_tn += us_tn
    tot_nt += us_nt
    print(f"{name:8s} N={N:6d} K={K:6d}: TN {us_tn:8.1f} us ({fl / us_tn / 1e6:6.0f} TFLOP/s)   transposes+NT {us_nt:8.1f} us "
          f"(NT GEMM alone {us_g:8.1f} us, {fl / us_g / 1e6:6.0f} TFLOP/s)")
print(f"layer total: TN {tot_tn:.0f} us vs {tot_nt:.0f} us")

# leading-dimension sensitivity of the TN kernel (row stride of dy / x in bytes vs the memory channel interleave)
print("qkv wgrad vs leading dimensions (elements): dy ld / x ld -> us")
N, K = 12288, 4096
for pd, px in ((0, 0), (64, 0), (128, 0), (320, 0), (0, 64), (64, 64), (320, 64)):
    dyw = torch.randn(Tn, N + pd, device=dev).to(torch.bfloat16)
    xw = torch.randn(Tn, K + px, device=dev).to(torch.bfloat16)
    out = torch.empty(N, K, dtype=torch.float32, device=dev)
    op = T.gemm_tn(dyw[:, :N], xw[:, :K], out, workspace=ws, run=False)
    print(f"   {N + pd:6d} / {K + px:5d}: {timeit(op.run):8.1f}")
