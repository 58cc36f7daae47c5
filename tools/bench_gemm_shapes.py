"""Time bl_gemm_bf16 on arbitrary shapes: python tools/bench_gemm_shapes.py M,N,K[,epi] ...   (epi: none|bias|gelu|res|swiglu)"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops
dev = torch.device("cuda:0")
EPI = {"none": ops.EPI_NONE, "bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "res": ops.EPI_BIAS_RES, "swiglu": ops.EPI_SWIGLU}
for spec in sys.argv[1:]:
    parts = spec.split(",")
    M, N, K = map(int, parts[:3])
    epi = EPI[parts[3]] if len(parts) > 3 else ops.EPI_NONE
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    out = torch.zeros(M, N // 2 if epi == ops.EPI_SWIGLU else N, device=dev, dtype=torch.bfloat16)
    kw = {}
    if epi not in (ops.EPI_NONE, ops.EPI_SWIGLU): kw["bias"] = torch.randn(N, device=dev).to(torch.bfloat16)
    if epi == ops.EPI_BIAS_RES: kw["res"] = torch.randn(M, N, device=dev).to(torch.bfloat16)
    op = ops.gemm(a, w, out, epi, run=False, **kw)
    for _ in range(5): op.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): op.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"M={M:5d} N={N:5d} K={K:5d} {parts[3] if len(parts) > 3 else 'none':5s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
