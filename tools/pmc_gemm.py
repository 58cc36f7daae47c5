"""Tiny driver for PMC passes: each Llama-2-7B prefill GEMM shape (B=16, S=288) launched 3 times.
rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_gemm.py   (and a second pass with WRITE_SIZE)
`python3 tools/pmc_gemm.py 9472` uses the training step's token count (B=32, S=296) instead."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 288
for name, N, K, epi in [("qkv", 12288, 4096, ops.EPI_NONE), ("o", 4096, 4096, ops.EPI_RES),
                        ("gate_up", 22016, 4096, ops.EPI_SWIGLU), ("down", 4096, 11008, ops.EPI_RES)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    out = torch.zeros(M, N // 2 if epi == ops.EPI_SWIGLU else N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out, epi, res=res if epi == ops.EPI_RES else None)
torch.cuda.synchronize()
