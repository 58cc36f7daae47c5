"""bl_adamw_f32 on one decoder layer's parameters (202 M fp32 master + moments + gradient, bf16 copy out): time, bytes/s,
checksum (A/B across library builds via BRIDGELANG_HIP_LIB).  python tools/bench_adamw.py"""
import sys, hashlib, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import train_ops as T, ops
dev = torch.device("cuda:0")
n = 4096 * 12288 + 4096 * 4096 + 22016 * 4096 + 4096 * 11008 + 2 * 4096 + 3     # + 3: a scalar tail
g = torch.Generator(device=dev).manual_seed(0)
p, m, v, gr = (torch.randn(n, device=dev, generator=g) * s for s in (0.02, 1e-4, 1e-8, 1e-3))
v = v.abs()
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
coef = torch.tensor([1.0, 0.7], device=dev)
op = T.adamw(p, m, v, gr, 3, 2e-5, weight_decay=0.01, norm_coef=coef, p_bf16=pb, run=False)
ops.run_all([op]); torch.cuda.synchronize()
h = hashlib.sha1(p.cpu().numpy().tobytes() + m.cpu().numpy().tobytes() + v.cpu().numpy().tobytes() + pb.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.run_all([op])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"adamw n = {n}: {us:7.1f} us  {30.0 * n / us / 1e6:5.2f} TB/s (30 B per parameter)  sha1 after one step {h}")
