"""The merged decode iteration's attention: six batches x 16 sequences x 32 heads in one launch, each with its own KV cache and
position (288 … 293), 32 layers' caches walked in one graph (cold caches). Time, KV bytes/s and a checksum (A/B across library
builds via BRIDGELANG_HIP_LIB).  python tools/bench_attn_decode_grouped.py"""
import sys, hashlib, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops
from bridgelang_amd.engine import rope_tables
dev = torch.device("cuda:0")
G, B, H, hd, L, cache_len = 6, 16, 32, 128, 32, 296
D = H * hd
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
kc = [[rnd(B, H, cache_len, hd) for _ in range(G)] for _ in range(L)]
vc = [[rnd(B, H, cache_len, hd) for _ in range(G)] for _ in range(L)]
qkv, o = rnd(G * B, 3 * D), torch.zeros(G * B, D, dtype=torch.bfloat16, device=dev)
cos, sin = rope_tables(hd, cache_len, 10000.0, dev)
pos = [288 + i for i in range(G)]
plan = [ops.attention_decode_rope_grouped(qkv, kc[l], vc[l], o, cos, sin, B=B, H=H, head_dim=hd, pos=pos, run=False) for l in range(L)]
ops.run_all(plan); torch.cuda.synchronize()
h = hashlib.sha1(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    ops.run_all(plan)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    gr.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 / L * 1e3
nb = sum(G * B * H * (p + 1) * hd * 2 * 2 for p in [288] * 1) / 1 * 1.0
nb = sum(B * H * (p + 1) * hd * 2 * 2 for p in pos)
print(f"grouped decode attention, {G} x {B} x {H} heads, Skv 289..294: {us:7.1f} us per layer  {nb / us / 1e6:5.2f} TB/s of K/V  sha1(o) {h}")
