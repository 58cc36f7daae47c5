"""Time bl_gemm_fp8 next to bl_gemm_bf16: python tools/bench_gemm_fp8.py M,N,K ..."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops
dev = torch.device("cuda:0")


def timeit(op, n=30):
    for _ in range(5): op.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): op.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for spec in sys.argv[1:]:
    M, N, K = map(int, spec.split(","))
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    t16 = timeit(ops.gemm(a, ops.pack_weight(w), out, ops.EPI_NONE, run=False))
    A8, sa, qop = ops.quantize_rows_fp8(a)
    W8, sw = ops.quantize_weight_fp8(w)
    t8 = timeit(ops.gemm_fp8(A8, sa, W8, sw, out, ops.EPI_NONE, run=False))
    tq = timeit(qop)
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d}: bf16 {t16:7.1f} us {fl / t16 / 1e6:7.1f} TF/s | fp8 {t8:7.1f} us {fl / t8 / 1e6:7.1f} TF/s "
          f"({t16 / t8:.2f}x) | row quantiser {tq:5.1f} us ({3.0 * M * K / tq / 1e3:.0f} GB/s)")
