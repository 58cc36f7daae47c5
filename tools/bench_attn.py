"""Micro-benchmark of the prefill / ViT attention kernel at the OpenVLA shapes (B = 16)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops
dev = torch.device("cuda:0")
for name, H, hd, S, causal in [("llama", 32, 128, 288, True), ("dino", 16, 64, 261, False), ("siglip", 16, 72, 256, False)]:
    B, D = 16, H * hd
    qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.zeros(B * S, D, device=dev, dtype=torch.bfloat16)
    st = (S * 3 * D, hd, 3 * D)
    op = ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], o, B=B, H=H, Sq=S, Skv=S, head_dim=hd, q_strides=st, k_strides=st,
                       v_strides=st, o_strides=(S * D, hd, D), causal=causal, run=False)
    for _ in range(3): op.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): op.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    print(f"{name:7s} H={H} hd={hd} S={S}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
# the fused form the Llama prefill uses: RoPE + KV-cache write inside the q / k / v loads (bl_attention_rope_bf16)
from bridgelang_amd.engine import rope_tables
B, H, hd, S = 16, 32, 128, 288
D = H * hd
cache_len = 320
qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
o = torch.zeros(B * S, D, device=dev, dtype=torch.bfloat16)
kc, vc = (torch.zeros(B, H, cache_len, hd, device=dev, dtype=torch.bfloat16) for _ in range(2))
cos, sin = rope_tables(hd, 2048, 10000.0, dev)
op = ops.attention_rope(qkv, kc, vc, o, cos, sin, B=B, S=S, H=H, head_dim=hd, pos0=0, key_mask=None, run=False)
for _ in range(3): op.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): op.run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"llama+rope H={H} hd={hd} S={S}: {us:7.1f} us  {4.0 * B * H * S * S * hd * 0.5 / us / 1e6:6.1f} TFLOP/s")
