"""GPU parity for the fused decode-step kernels: RMSNorm fused into the weight-streaming GEMM, RoPE + KV-cache append
fused into the single-query attention. Checked against the oracle AND against the unfused HIP kernels."""
import pytest
import torch

from conftest import rand_bf16
from oracle import restate as R
from test_ops_gpu import close_bf16, dv, pk

pytestmark = pytest.mark.gpu
P = R.Prec(True)


@pytest.mark.parametrize("M,N,K", [(16, 1024, 4096), (3, 3072, 512), (16, 512, 1536)])
def test_skinny_gemm_with_fused_rmsnorm(dev, M, N, K):
    from bridgelang_amd import ops
    x, g, w = rand_bf16((M, K), 1, 3.0), P.rb(rand_bf16((K,), 2, 0.02) + 1), rand_bf16((N, K), 3, 0.05)
    X, G, W = dv(x, dev), dv(g, dev), pk(w, dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(X, W, out, ops.EPI_NONE, a_norm=(G, 1e-6))
    ref = R.linear(P, R.rmsnorm(P, x, g, 1e-6), w)
    close_bf16(out, ref, "fused rmsnorm + skinny gemm", min_exact=0.95)
    # vs the unfused HIP pair
    h = torch.empty_like(X)
    ops.rmsnorm(X, G, h, 1e-6)
    out2 = torch.empty_like(out)
    ops.gemm(h, W, out2, ops.EPI_NONE, skinny=True)
    close_bf16(out, out2.cpu().float(), "fused vs unfused", min_exact=0.97)
    # SwiGLU + fused norm (decode gate/up)
    I = N // 2
    gu = torch.stack([w[:I], w[I:]], 1).reshape(N, K)
    out3 = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    ops.gemm(X, pk(gu, dev), out3, ops.EPI_SWIGLU, a_norm=(G, 1e-6))
    hn = R.rmsnorm(P, x, g, 1e-6)
    ga, up = R.linear(P, hn, w[:I]), R.linear(P, hn, w[I:])
    close_bf16(out3, P.rb(P.rb(torch.nn.functional.silu(ga)) * up), "fused rmsnorm + swiglu", min_exact=0.93)


@pytest.mark.parametrize("pos", [0, 5, 288, 293])
def test_attention_decode_rope(dev, pos):
    from bridgelang_amd import ops
    B, H, hd, cache_len = 3, 4, 128, 320
    D = H * hd
    qkv = rand_bf16((B, 3 * D), 1 + pos)
    kc, vc = rand_bf16((B, H, cache_len, hd), 2), rand_bf16((B, H, cache_len, hd), 3)
    cos, sin = R.rope_tables(hd, 512, 10000.0)
    Q, KC, VC = dv(qkv, dev), dv(kc, dev), dv(vc, dev)
    o = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
    ops.attention_decode_rope(Q, KC, VC, o, dv(cos, dev), dv(sin, dev), B=B, H=H, head_dim=hd, pos=pos)
    # oracle: rotate q,k at pos; append; attend over 0..pos
    t = qkv.view(B, 1, 3, H, hd).permute(2, 0, 3, 1, 4)          # [3, B, H, 1, hd]
    q_r, k_r = R.apply_rope(P, t[0], cos, sin, pos), R.apply_rope(P, t[1], cos, sin, pos)
    k_all = torch.cat([kc[:, :, :pos], k_r], dim=2)
    v_all = torch.cat([vc[:, :, :pos], t[2]], dim=2)
    ref = R.attention(P, q_r, k_all, v_all, hd ** -0.5, False).reshape(B, D)
    close_bf16(o, ref, "decode attention + rope", rtol=2 ** -5, atol_scale=2 ** -7, min_exact=0.55)
    # cache row `pos` appended bit-exactly, everything else untouched
    assert torch.equal(KC.cpu().float()[:, :, pos], k_r[:, :, 0]) and torch.equal(VC.cpu().float()[:, :, pos], t[2][:, :, 0])
    keep = [i for i in range(cache_len) if i != pos]
    assert torch.equal(KC.cpu().float()[:, :, keep], kc[:, :, keep]) and torch.equal(VC.cpu().float()[:, :, keep], vc[:, :, keep])
    # vs the unfused HIP pair (rope_kvcache + attention_decode): identical bits
    Q2, KC2, VC2 = dv(qkv, dev), dv(kc, dev), dv(vc, dev)
    o2 = torch.zeros_like(o)
    ops.rope_kvcache(Q2, dv(cos, dev), dv(sin, dev), KC2, VC2, B=B, S=1, H=H, head_dim=hd, pos0=pos)
    cs = (H * cache_len * hd, cache_len * hd, hd)
    ops.attention_decode(Q2, KC2, VC2, o2, B=B, H=H, Skv=pos + 1, head_dim=hd, q_strides=(3 * D, hd, 3 * D),
                         k_strides=cs, v_strides=cs, o_strides=(D, hd, D))
    assert torch.equal(KC.cpu(), KC2.cpu()) and torch.equal(VC.cpu(), VC2.cpu())
    close_bf16(o, o2.cpu().float(), "fused vs unfused decode attention", rtol=2 ** -6, min_exact=0.9)
