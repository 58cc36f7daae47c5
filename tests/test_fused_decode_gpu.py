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


@pytest.mark.parametrize("M,N,K", [(96, 4096, 4096), (96, 1024, 11008), (17, 512, 512), (128, 3072, 1536), (40, 2048, 1024),
                                   (112, 640, 5120), (96, 256, 13824), (33, 22016, 512),
                                   # the wide layers' weight-streaming form (gemm_rows_stream_kernel): 6-wave workgroups walking
                                   # all of K (gate/up), two K-halves + tree reduce (qkv), 8-wave workgroups (lm_head width),
                                   # ragged row counts, a surplus wave in the last workgroup (6176 / 16 = 386 tiles)
                                   (96, 22016, 4096), (70, 12288, 4096), (96, 32064, 1024), (16, 6176, 5120), (1, 8192, 1024)])
def test_skinny_rows_bit_identical_to_skinny(dev, M, N, K):
    """bl_gemm_skinny_rows_bf16 (up to 128 stacked rows, the merged decode iteration) must give every row EXACTLY what
    bl_gemm_skinny_bf16 gives that row in a batch of <= 16 — same 8-way K partition, per-slice k order and combine order
    — for every epilogue the decode path uses; and against the oracle to bf16 rounding."""
    from bridgelang_amd import ops
    a, w, r = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((M, N), 4)
    A, W, Rr = dv(a, dev), pk(w, dev), dv(r, dev)
    I = N // 2
    Wgu = pk(torch.stack([w[:I], w[I:]], 1).reshape(N, K), dev)
    cases = [("NONE", ops.EPI_NONE, W, torch.bfloat16, N, None), ("RES", ops.EPI_RES, W, torch.bfloat16, N, Rr),
             ("SWIGLU", ops.EPI_SWIGLU, Wgu, torch.bfloat16, I, None), ("F32_BF16R", ops.EPI_F32_BF16R, W, torch.float32, N, None),
             ("F32", ops.EPI_F32, W, torch.float32, N, None)]
    ws = torch.empty(4 * M * N * 4, dtype=torch.uint8, device=dev)
    for name, epi, Wp, dt, n_out, res in cases:
        got = torch.full((M, n_out), 7.0, dtype=dt, device=dev)
        ops.gemm(A, Wp, got, epi, res=res, skinny_rows=True)              # one workgroup walks all 8 K-slices
        got4 = torch.full((M, n_out), 7.0, dtype=dt, device=dev)
        ops.gemm(A, Wp, got4, epi, res=res, skinny_rows=True, workspace=ws)   # narrow N: K split over 4 workgroups + tree reduce
        assert torch.equal(got, got4), f"{name}: the K split across workgroups changed {(got != got4).sum().item()} elements"
        want = torch.empty_like(got)
        for r0 in range(0, M, 16):      # the per-batch kernel, 16 rows (one batch) at a time; odd group sizes too
            r1 = min(M, r0 + (16 if r0 % 32 == 0 else 5))
            for q0, q1 in ((r0, r1), (r1, min(M, r0 + 16))):
                if q1 > q0:
                    ops.gemm(A[q0:q1], Wp, want[q0:q1], epi, res=None if res is None else res[q0:q1], skinny=True)
        assert torch.equal(got, want), f"{name}: {(got != want).sum().item()} of {got.numel()} elements differ"
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, out, ops.EPI_NONE, skinny_rows=True)
    close_bf16(out, R.linear(P, a, w), "skinny rows vs oracle")


@pytest.mark.parametrize("M,N,K", [(96, 12288, 4096), (48, 22016, 1024), (96, 32064, 1024)])
def test_rows_stream_equals_mid_kernel(dev, M, N, K, monkeypatch):
    """The wide layers of the merged decode iteration run on gemm_rows_stream_kernel (weights HBM → VGPR, rows staged once
    per workgroup); BL_ROWS_STREAM=0 keeps them on gemm_mid_kernel<SK>. Both walk the skinny order: equal bit for bit,
    with and without the K split over two workgroups (workspace given / not given)."""
    from bridgelang_amd import ops
    a, w = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05)
    A, W = dv(a, dev), pk(w, dev)
    ws = torch.empty(4 * M * N * 4, dtype=torch.uint8, device=dev)
    got = {}
    for on in ("0", "1"):
        monkeypatch.setenv("BL_ROWS_STREAM", on)
        for wsp in (None, ws):
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, out, ops.EPI_NONE, skinny_rows=True, workspace=wsp)
            got[on, wsp is None] = out.cpu()
    assert torch.equal(got["0", True], got["1", True]) and torch.equal(got["0", False], got["1", False])
    assert torch.equal(got["1", True], got["1", False])


@pytest.mark.parametrize("rows,K", [(96, 4096), (16, 512), (33, 1536), (5, 11008), (50, 5120), (128, 1024), (17, 13824)])
def test_rmsnorm_skinny_equals_fused_norm(dev, rows, K):
    """bl_rmsnorm_skinny_bf16 + a GEMM == the skinny GEMM's fused a_norm, bit for bit (probed through an identity-like
    weight: y = x_norm · Iᵀ returns the normalised activations exactly), and within bf16 rounding of the oracle."""
    from bridgelang_amd import ops
    x, g = rand_bf16((rows, K), 1, 3.0), P.rb(rand_bf16((K,), 2, 0.02) + 1)
    X, G = dv(x, dev), dv(g, dev)
    h = torch.empty_like(X)
    ops.rmsnorm_skinny(X, G, h, 1e-6)
    close_bf16(h, R.rmsnorm(P, x, g, 1e-6), "rmsnorm_skinny vs oracle")
    n_probe = 512
    cols = torch.randperm(K, generator=torch.Generator().manual_seed(3))[:n_probe]
    eye = torch.zeros(n_probe, K)
    eye[torch.arange(n_probe), cols] = 1.0
    Wp = pk(eye, dev)
    w2 = pk(rand_bf16((256, K), 5, 0.05), dev)
    for r0 in range(0, rows, 16):
        r1 = min(rows, r0 + 16)
        fused = torch.empty(r1 - r0, n_probe, dtype=torch.bfloat16, device=dev)
        ops.gemm(X[r0:r1], Wp, fused, ops.EPI_NONE, a_norm=(G, 1e-6))
        assert torch.equal(fused, h[r0:r1][:, cols.to(dev)]), "normalised activations differ from the fused norm's"
        a = torch.empty(r1 - r0, 256, dtype=torch.bfloat16, device=dev)
        b = torch.empty_like(a)
        ops.gemm(X[r0:r1], w2, a, ops.EPI_NONE, a_norm=(G, 1e-6))
        ops.gemm(h[r0:r1], w2, b, ops.EPI_NONE, skinny=True)
        assert torch.equal(a, b)
