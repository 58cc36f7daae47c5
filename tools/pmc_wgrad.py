"""Tiny driver for PMC passes on the weight-gradient TN GEMM (Llama-2-7B layer shapes, T = 32 x 296 tokens), three launches
per shape:  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/pmc_wgrad.py   (and a pass with
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import train_ops as T

dev = torch.device("cuda:0")
Tn = 9472
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for name, N, K in [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]:
    dy = torch.randn(Tn, N, device=dev).to(torch.bfloat16)
    x = torch.randn(Tn, K, device=dev).to(torch.bfloat16)
    out = torch.empty(N, K, dtype=torch.float32, device=dev)
    for _ in range(3):
        T.gemm_tn(dy, x, out, workspace=ws)
torch.cuda.synchronize()
