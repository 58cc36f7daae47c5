"""Micro-benchmark of the weight-streaming (M<=16) GEMM on the Llama-2-7B decode shapes; prints achieved HBM GB/s."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops

dev = torch.device("cuda:0")
shapes = [("qkv", 12288, 4096, ops.EPI_NONE), ("o", 4096, 4096, ops.EPI_RES), ("gate_up", 22016, 4096, ops.EPI_SWIGLU),
          ("down", 4096, 11008, ops.EPI_RES), ("lm_head", 32064, 4096, ops.EPI_F32_BF16R)]
M = 16
for name, N, K, epi in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    # 8 distinct weight copies so the stream always comes from HBM (total > 256 MiB Infinity Cache)
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(ncopy)]
    nout = N // 2 if epi == ops.EPI_SWIGLU else N
    out = torch.zeros(M, nout, device=dev, dtype=torch.float32 if epi == ops.EPI_F32_BF16R else torch.bfloat16)
    res = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    plan = [ops.gemm(x, w, out, epi, res=res if epi == ops.EPI_RES else None, skinny=True, run=False) for w in ws]
    for op in plan: op.run()
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for op in plan: op.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * len(plan))
    print(f"{name:8s} N={N:6d} K={K:6d}: {us:7.1f} us  {N * K * 2 / us / 1e3:7.1f} GB/s")
