"""Micro-benchmark of the tiled GEMM on the OpenVLA-7B prefill / ViT shapes (B=16). BL_GEMM_TILE=128|256 forces a kernel."""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops

dev = torch.device("cuda:0")
M_LLM, M_DINO, M_SIG = 16 * 288, 16 * 261, 16 * 256
shapes = [("llama qkv", M_LLM, 12288, 4096, ops.EPI_NONE), ("llama o", M_LLM, 4096, 4096, ops.EPI_RES),
          ("llama gate_up", M_LLM, 22016, 4096, ops.EPI_SWIGLU), ("llama down", M_LLM, 4096, 11008, ops.EPI_RES),
          ("dino qkv", M_DINO, 3072, 1024, ops.EPI_BIAS), ("dino proj", M_DINO, 1024, 1024, ops.EPI_BIAS_RES),
          ("dino fc1", M_DINO, 4096, 1024, ops.EPI_BIAS_GELU), ("dino fc2", M_DINO, 1024, 4096, ops.EPI_BIAS_RES),
          ("sig qkv", M_SIG, 3456, 1152, ops.EPI_BIAS), ("sig proj", M_SIG, 1152, 1152, ops.EPI_BIAS_RES),
          ("sig fc1", M_SIG, 4352, 1152, ops.EPI_BIAS_GELU), ("sig fc2", M_SIG, 1152, 4352, ops.EPI_BIAS_RES),
          ("proj fc1", M_SIG, 8704, 2176, ops.EPI_BIAS_GELU), ("proj fc2", M_SIG, 4096, 8704, ops.EPI_BIAS_GELU),
          ("square 8k", 8192, 8192, 8192, ops.EPI_NONE), ("square 4k", 4096, 4096, 4096, ops.EPI_NONE)]
print("tile =", os.environ.get("BL_GEMM_TILE", "auto"))
tot = 0.0
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    nout = N // 2 if epi == ops.EPI_SWIGLU else N
    out = torch.zeros(M, nout, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU, ops.EPI_BIAS_RES): kw["bias"] = bias
    if epi in (ops.EPI_RES, ops.EPI_BIAS_RES): kw["res"] = res
    op = ops.gemm(a, w, out, epi, run=False, workspace=ws, **kw)
    for _ in range(3): op.run()
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): op.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    if name.startswith("llama"): tot += us
    print(f"{name:14s} M={M:5d} N={N:6d} K={K:6d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
print(f"llama layer GEMMs total: {tot:.1f} us")
