"""Micro-benchmark of the attention forward (lse form) + backward pair at the training shapes (B = 32: Llama S = 296, ViT towers)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import train_ops as T
dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, H, hd, S, causal in [("llama", 32, 128, 296, True), ("dino", 16, 64, 261, False), ("siglip", 16, 72, 256, False),
                               ("llama-long", 32, 128, 600, True)]:
    B = 32 if S < 400 else 8
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.zeros(B * S, D, device=dev, dtype=torch.bfloat16)
    g = torch.randn(B * S, D, device=dev).to(torch.bfloat16)
    pad = (S + 31) // 32 * 32
    lse, delta = torch.zeros(B * H * pad, device=dev), torch.zeros(B * H * pad, device=dev)
    dqkv = torch.zeros_like(qkv)
    st, so = (S * 3 * D, hd, 3 * D), (S * D, hd, D)
    kw = dict(B=B, H=H, Sq=S, Skv=S, head_dim=hd, q_strides=st, k_strides=st, v_strides=st, o_strides=so, causal=causal)
    f = T.attention_lse(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, run=False, **kw)
    b = T.attention_backward(qkv, qkv[:, D:], qkv[:, 2 * D:], o, g, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], run=False, **kw)
    tf, tb = timed(f.run), timed(b.run)
    fl = 2.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    print(f"{name:10s} B={B} H={H} hd={hd} S={S}: forward {tf:7.1f} us ({2 * fl / tf / 1e6:5.0f} TFLOP/s)  backward (dq + dk/dv) {tb:7.1f} us "
          f"({7 * fl / tb / 1e6:5.0f} TFLOP/s executed, {5 * fl / tb / 1e6:5.0f} algorithmic)")
