"""RMSNorm backward at the 7B training shape (9472 rows x 4096): time and a checksum of dx / dw (A/B across builds or
BL_NORM_BWD_DRL=0/1: the LDS path of dres must not change a bit).  python tools/bench_norm_bwd.py [rows] [dim]"""
import sys, hashlib, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import train_ops as T, ops
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 9472
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ln = len(sys.argv) > 3 and sys.argv[3] == "ln"        # LayerNorm backward (ViT towers) instead of RMSNorm
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(dev)
x, dy, dres, w = mk(rows, dim), mk(rows, dim), mk(rows, dim), (torch.randn(dim, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
dx = torch.empty_like(x); dw = torch.empty(dim, dtype=torch.float32, device=dev)
ws = torch.empty(((rows + 15) // 16) * dim * 2, dtype=torch.float32, device=dev)
db = torch.empty(dim, dtype=torch.float32, device=dev)
for tag, dr in (("with dres", dres), ("no dres  ", None)):
    op = (T.layernorm_backward(x, w, dy, dx, dw, db, ws, 1e-6, dres=dr, run=False) if ln else
          T.rmsnorm_backward(x, w, dy, dx, dw, ws, 1e-5, dres=dr, run=False))
    ops.run_all([op]); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(20):
            ops.run_all([op])
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gph.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    nb = rows * dim * 2 * (4 if dr is not None else 3)
    h = hashlib.sha1(dx.cpu().view(torch.int16).numpy().tobytes() + dw.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{'layernorm' if ln else 'rmsnorm'} backward {rows} x {dim} {tag}: {us:7.1f} us  {nb / us / 1e6:6.2f} TB/s  sha1(dx,dw) {h}")
