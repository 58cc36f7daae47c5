"""GPU parity of the backward / optimizer kernels against torch.autograd over the CPU oracle's forward functions
(oracle/restate.py in bf16-rounding mode: casts are identity in the backward) and against torch.optim.AdamW."""
import numpy as np
import pytest
import torch

from conftest import rand_bf16
from oracle import restate as R
from test_ops_gpu import close_bf16, dv

pytestmark = pytest.mark.gpu
P = R.Prec(True)


def grad_close(got, ref, what, tol=2e-2):
    got, ref = got.float().cpu(), ref.float()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


def test_cross_entropy_backward(dev):
    from bridgelang_amd import ops, train_ops as T
    rows, V = 24, 32064
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(rows, V, generator=g) * 2).to(torch.bfloat16).float()
    tgt = torch.randint(0, V, (rows,), generator=g)
    tgt[::3] = -100
    lg = logits.clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lg, tgt, ignore_index=-100)
    loss.backward()
    L, Tg = logits.to(dev), tgt.to(dev)
    row_loss, mc = torch.empty(rows, device=dev), torch.empty(2, device=dev)
    ops.cross_entropy(L, Tg, row_loss, mc)
    assert abs(mc[0].item() - loss.item()) < 1e-4 * abs(loss.item())
    dl = torch.empty(rows, V, dtype=torch.bfloat16, device=dev)
    T.cross_entropy_backward(L, Tg, mc, dl)
    grad_close(dl, lg.grad, "dlogits", 1e-2)
    assert (dl.cpu()[::3] == 0).all()


@pytest.mark.parametrize("rows,dim", [(70, 512), (300, 4096)])
def test_rmsnorm_backward(dev, rows, dim):
    from bridgelang_amd import train_ops as T
    x, w, dy, dres = rand_bf16((rows, dim), 1, 2.0), P.rb(rand_bf16((dim,), 2, 0.02) + 1), rand_bf16((rows, dim), 3), rand_bf16((rows, dim), 4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = R.rmsnorm(P, xr, wr, 1e-6)
    (y * dy).sum().backward()
    dx = torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)
    dw = torch.empty(dim, device=dev)
    ws = torch.empty(((rows + 63) // 64) * dim, device=dev)
    T.rmsnorm_backward(dv(x, dev), dv(w, dev), dv(dy, dev), dx, dw, ws, 1e-6, dres=dv(dres, dev))
    grad_close(dx, xr.grad + dres, "rmsnorm dx (+residual)")
    grad_close(dw, wr.grad, "rmsnorm dw")
    T.rmsnorm_backward(dv(x, dev), dv(w, dev), dv(dy, dev), dx, dw, ws, 1e-6)
    grad_close(dx, xr.grad, "rmsnorm dx")


@pytest.mark.parametrize("rows,dim", [(333, 4096), (64, 5120), (700, 2560)])
def test_rmsnorm_backward_dres_through_lds_is_bit_identical(dev, rows, dim, monkeypatch):
    """The decoder's RMSNorm backward carries the residual-stream gradient row through LDS (LDS-DMA under the reduction
    passes) instead of loading it inside the output pass: same arithmetic, so dx and dw must equal the plain form
    (BL_NORM_BWD_DRL=0) bit for bit — ragged last block, 13B width (one workgroup per CU), 2560 = five whole chunks."""
    from bridgelang_amd import train_ops as T
    x, w, dy, dres = rand_bf16((rows, dim), 1, 2.0), P.rb(rand_bf16((dim,), 2, 0.02) + 1), rand_bf16((rows, dim), 3), rand_bf16((rows, dim), 4)
    got = {}
    for on in ("0", "1"):
        monkeypatch.setenv("BL_NORM_BWD_DRL", on)
        dx = torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)
        dw = torch.empty(dim, device=dev)
        ws = torch.empty(((rows + 15) // 16) * dim, device=dev)
        T.rmsnorm_backward(dv(x, dev), dv(w, dev), dv(dy, dev), dx, dw, ws, 1e-6, dres=dv(dres, dev))
        got[on] = (dx.cpu(), dw.cpu())
    assert torch.equal(got["0"][0], got["1"][0]) and torch.equal(got["0"][1], got["1"][1])


def test_swiglu_gelu_rope_backward(dev):
    from bridgelang_amd import train_ops as T
    M, I = 37, 1536
    gu, dact = rand_bf16((M, 2 * I), 1, 1.5), rand_bf16((M, I), 2)
    gur = gu.clone().requires_grad_(True)
    act = P.rb(P.rb(torch.nn.functional.silu(gur[:, 0::2])) * gur[:, 1::2])
    (act * dact).sum().backward()
    A = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    T.swiglu(dv(gu, dev), A)
    close_bf16(A, act.detach(), "swiglu fwd")
    dgu = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=dev)
    T.swiglu_backward(dv(gu, dev), dv(dact, dev), dgu)
    grad_close(dgu, gur.grad, "swiglu bwd")
    # GELU
    x, dy = rand_bf16((M, 832), 3, 2.0), rand_bf16((M, 832), 4)
    xr = x.clone().requires_grad_(True)
    y = R.gelu(P, xr)
    (y * dy).sum().backward()
    Y = torch.empty(M, 832, dtype=torch.bfloat16, device=dev)
    T.gelu(dv(x, dev), Y)
    close_bf16(Y, y.detach(), "gelu fwd")
    dx = torch.empty_like(Y)
    T.gelu_backward(dv(x, dev), dv(dy, dev), dx)
    grad_close(dx, xr.grad, "gelu bwd")
    # RoPE
    B, S, H, hd = 2, 19, 4, 128
    D = H * hd
    qkv, dqkv = rand_bf16((B * S, 3 * D), 5), rand_bf16((B * S, 3 * D), 6)
    cos, sin = R.rope_tables(hd, 64, 10000.0)
    qr = qkv.clone().requires_grad_(True)
    t = qr.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    out = torch.stack([R.apply_rope(P, t[0], cos, sin, 0), R.apply_rope(P, t[1], cos, sin, 0), t[2]])
    (out * dqkv.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)).sum().backward()
    G = dv(dqkv, dev)
    T.rope_backward(G, dv(cos, dev), dv(sin, dev), B=B, S=S, H=H, head_dim=hd)
    grad_close(G, qr.grad, "rope bwd")


def test_transpose_colsum_wgrad(dev):
    """wgrad dW = dY^T X as an NT GEMM over transposed, zero-padded, packed operands; bias grad = column sums."""
    from bridgelang_amd import ops, train_ops as T
    M, N, K = 300, 192, 256
    dy, x = rand_bf16((M, N), 1), rand_bf16((M, K), 2)
    Mp = (M + 63) // 64 * 64
    dyT = torch.empty(N, Mp, dtype=torch.bfloat16, device=dev)
    xT = torch.empty(K, Mp, dtype=torch.bfloat16, device=dev)
    T.transpose_pad(dv(dy, dev), dyT, Mp)
    T.transpose_pad(dv(x, dev), xT, Mp)
    assert torch.equal(dyT.cpu().float()[:, :M], dy.t()) and (dyT.cpu()[:, M:] == 0).all()
    dW = torch.empty(N, K, dtype=torch.float32, device=dev)
    ops.gemm(dyT, ops.pack_weight(xT), dW, ops.EPI_F32)
    ref = dy.t() @ x
    assert torch.allclose(dW.cpu(), ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item())
    out, ws = torch.empty(N, device=dev), torch.empty(((M + 255) // 256) * N, device=dev)
    T.colsum(dv(dy, dev), out, ws)
    assert torch.allclose(out.cpu(), dy.sum(0), rtol=1e-5, atol=1e-4)


def test_adamw_and_clip_match_torch(dev):
    from bridgelang_amd import train_ops as T
    g = torch.Generator().manual_seed(0)
    shapes = [(1000,), (64, 129), (7,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref_p = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.AdamW(ref_p, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    dp = [p.clone().to(dev) for p in ps]
    ms, vs = [torch.zeros_like(p) for p in dp], [torch.zeros_like(p) for p in dp]
    for step in range(1, 4):
        grads = [torch.randn(s, generator=g) * (10.0 if step == 2 else 0.1) for s in shapes]
        for p, gr in zip(ref_p, grads):
            p.grad = gr.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref_p, max_norm=1.0)
        opt.step()
        dg = [gr.to(dev) for gr in grads]
        partial = torch.zeros(3 * 8, device=dev)
        for i, gr in enumerate(dg):
            T.sumsq_partial(gr.view(-1), partial[i * 8:(i + 1) * 8])
        nc = torch.empty(2, device=dev)
        T.clip_coef(partial, 1.0, nc)
        assert abs(nc[0].item() - norm_ref.item()) <= 1e-5 * norm_ref.item()
        for p, m, v, gr in zip(dp, ms, vs, dg):
            bf = torch.empty(p.shape, dtype=torch.bfloat16, device=dev)
            T.adamw(p.view(-1), m.view(-1), v.view(-1), gr.view(-1), step, 2e-5, weight_decay=0.01, norm_coef=nc, p_bf16=bf)
            assert torch.equal(bf.float(), p.to(torch.bfloat16).float())
        for p, r in zip(dp, ref_p):
            assert torch.allclose(p.cpu(), r.detach(), rtol=1e-6, atol=1e-7), (p.cpu() - r.detach()).abs().max()


def test_embed_backward(dev):
    from bridgelang_amd import train_ops as T
    B, L, D, V, NP = 2, 6, 64, 50, 4
    ids = torch.tensor([[1, 5, 7, 5, 9, 2], [1, 5, 3, 3, 3, 2]])
    dx = rand_bf16((B, L + NP, D), 1)
    dw = torch.zeros(V, D, device=dev)
    T.embed_backward(ids.to(dev), dv(dx, dev), dw, NP)
    ref = torch.zeros(V, D)
    for b in range(B):
        for j in range(L):
            ref[ids[b, j]] += dx[b, 0 if j == 0 else j + NP]
    assert torch.allclose(dw.cpu(), ref, atol=1e-5)


@pytest.mark.parametrize("hd,H,S,causal,masked", [(128, 4, 290, True, False), (128, 2, 100, True, True), (128, 2, 33, True, False),
                                                  (64, 4, 257, False, False), (72, 2, 256, False, False),
                                                  # beyond 320 positions: the chunked kernels (prompts of 56+ tokens)
                                                  (128, 2, 406, True, False), (128, 2, 600, True, True), (128, 1, 1030, True, False),
                                                  (64, 2, 406, False, False), (72, 2, 600, False, True)])
def test_attention_backward(dev, hd, H, S, causal, masked):
    """dQ/dK/dV of the whole-sequence attention vs autograd over the oracle's attention (fused qkv-row strides, so the
    grads land in the three thirds of one dqkv buffer as the training step uses them)."""
    from bridgelang_amd import train_ops as T
    B, D = 3, H * hd
    qkv, dout = rand_bf16((B * S, 3 * D), hd + S), rand_bf16((B * S, D), 7)
    mask = None
    lens = [S] * B
    if masked:
        lens = [S, 63 if S <= 320 else S - 137, 17]
        mask = torch.zeros(B, S, dtype=torch.uint8)
        for i, n in enumerate(lens):
            mask[i, :n] = 1
    valid = torch.zeros(B, S, 1, 1)
    for i, n in enumerate(lens):
        valid[i, :n] = 1
    qr = qkv.clone().requires_grad_(True)
    t = qr.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref_o = R.attention(P, t[0], t[1], t[2], hd ** -0.5, causal, key_mask=mask).permute(0, 2, 1, 3)   # [B,S,H,hd]
    g_o = dout.view(B, S, H, hd) * valid           # padded query rows carry no gradient (their loss labels are -100)
    (ref_o * g_o).sum().backward()
    Q, G = dv(qkv, dev), dv(g_o.reshape(B * S, D), dev)
    o = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
    pad = (S + 31) // 32 * 32
    lse = torch.full((B * H * pad,), float("nan"), device=dev)
    delta = torch.full((B * H * pad,), float("nan"), device=dev)
    st, so = (S * 3 * D, hd, 3 * D), (S * D, hd, D)
    kw = dict(B=B, H=H, Sq=S, Skv=S, head_dim=hd, q_strides=st, k_strides=st, v_strides=st, o_strides=so, causal=causal,
              key_mask=mask.to(dev) if masked else None)
    T.attention_lse(Q, Q[:, D:], Q[:, 2 * D:], o, lse, **kw)
    got_o = o.cpu().float().view(B, S, H, hd)
    for i, n in enumerate(lens):
        close_bf16(got_o[i, :n], ref_o.detach()[i, :n], "attention fwd (lse variant)", rtol=2 ** -5, atol_scale=2 ** -7, min_exact=0.55)
    # lse against the oracle's scores
    with torch.no_grad():
        s = (t[0] @ t[1].transpose(-1, -2)) * hd ** -0.5
        if causal:
            s = s.masked_fill(torch.ones(S, S, dtype=torch.bool).triu(1), float("-inf"))
        if masked:
            s = s.masked_fill(~mask.bool().view(B, 1, 1, S), float("-inf"))
        lse_ref = torch.logsumexp(s, -1) / np.log(2.0)
    lse_got = lse.cpu().view(B, H, pad)[:, :, :S]
    for i, n in enumerate(lens):
        assert torch.allclose(lse_got[i, :, :n], lse_ref[i, :, :n], rtol=1e-4, atol=1e-3)
    dqkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device=dev)
    T.attention_backward(Q, Q[:, D:], Q[:, 2 * D:], o, G, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], **kw)
    got, ref = dqkv.cpu().float().view(B, S, 3, H, hd), qr.grad.view(B, S, 3, H, hd)
    for i, n in enumerate(lens):
        for j, nm in enumerate("qkv"):
            grad_close(got[i, :n, j], ref[i, :n, j], f"d{nm} b={i}", tol=2.5e-2)


@pytest.mark.parametrize("hd,H,S,causal", [(128, 3, 290, True), (128, 2, 37, True), (64, 2, 257, False), (72, 2, 300, False)])
def test_attention_backward_chunked_equals_whole_sequence(dev, hd, H, S, causal, monkeypatch):
    """The long-sequence (chunked) backward kernels keep the whole-sequence kernels' summation order: forced onto a short
    sequence (BL_ATTN_BWD_CHUNKED) they must reproduce dq / dk / dv and delta bit for bit."""
    from bridgelang_amd import train_ops as T
    B, D = 2, H * hd
    Q, G = dv(rand_bf16((B * S, 3 * D), 11), dev), dv(rand_bf16((B * S, D), 12), dev)
    mask = torch.ones(B, S, dtype=torch.uint8)
    mask[1, S - 9:] = 0
    pad = (S + 31) // 32 * 32
    o = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B * H * pad, device=dev)
    st, so = (S * 3 * D, hd, 3 * D), (S * D, hd, D)
    kw = dict(B=B, H=H, Sq=S, Skv=S, head_dim=hd, q_strides=st, k_strides=st, v_strides=st, o_strides=so, causal=causal,
              key_mask=mask.to(dev))
    T.attention_lse(Q, Q[:, D:], Q[:, 2 * D:], o, lse, **kw)
    outs = []
    for chunked in (False, True):
        if chunked:
            monkeypatch.setenv("BL_ATTN_BWD_CHUNKED", "1")
        delta = torch.zeros(B * H * pad, device=dev)
        dqkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device=dev)
        T.attention_backward(Q, Q[:, D:], Q[:, 2 * D:], o, G, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], **kw)
        torch.cuda.synchronize()
        outs.append((dqkv.cpu(), delta.cpu()))
    monkeypatch.delenv("BL_ATTN_BWD_CHUNKED")
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("rows,dim", [(261 * 2, 256), (300, 1152)])
def test_layernorm_and_layerscale_backward(dev, rows, dim):
    from bridgelang_amd import train_ops as T
    x, w, b = rand_bf16((rows, dim), 1, 2.0), P.rb(rand_bf16((dim,), 2, 0.02) + 1), rand_bf16((dim,), 5, 0.02)
    dy, dres = rand_bf16((rows, dim), 3), rand_bf16((rows, dim), 4)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    (R.layernorm(P, xr, wr, br, 1e-6) * dy).sum().backward()
    dx = torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)
    dw, db = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
    ws = torch.empty(2 * ((rows + 15) // 16) * dim, device=dev)
    T.layernorm_backward(dv(x, dev), dv(w, dev), dv(dy, dev), dx, dw, db, ws, 1e-6, dres=dv(dres, dev))
    grad_close(dx, xr.grad + dres, "layernorm dx (+residual)")
    grad_close(dw, wr.grad, "layernorm dw")
    grad_close(db, br.grad, "layernorm db")
    # LayerScale residual: y = rb(rb(u * ls) + res)
    u, ls, res = rand_bf16((rows, dim), 6), P.rb(rand_bf16((dim,), 7, 0.02) + 0.1), rand_bf16((rows, dim), 8)
    ur, lr_ = u.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    y = P.rb(P.rb(ur * lr_) + res)
    (y * dy).sum().backward()
    Y = torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)
    T.scale_residual(dv(u, dev), dv(ls, dev), dv(res, dev), Y)
    assert torch.equal(Y.cpu().float(), y.detach())
    du, dls = torch.empty_like(Y), torch.empty(dim, device=dev)
    T.layerscale_backward(dv(dy, dev), dv(u, dev), dv(ls, dev), du, dls, torch.empty(((rows + 63) // 64) * dim, device=dev))
    grad_close(du, ur.grad, "layerscale du")
    grad_close(dls, lr_.grad, "layerscale dscale")


@pytest.mark.parametrize("rows,cols", [(300, 192), (4608, 4096), (37, 64), (1000, 1088)])
def test_fast_transpose_and_transpose_pack(dev, rows, cols):
    from bridgelang_amd import ops, train_ops as T
    a = rand_bf16((rows, cols), rows + cols)
    rp = (rows + 63) // 64 * 64
    A = dv(a, dev)
    t = torch.full((cols, rp), 7.0, dtype=torch.bfloat16, device=dev)
    T.transpose_pad(A, t, rp)
    assert torch.equal(t.cpu().float()[:, :rows], a.t()) and (t.cpu()[:, rows:] == 0).all()
    pk = torch.full((cols // 16, rp // 32, 64, 8), 7.0, dtype=torch.bfloat16, device=dev)
    T.transpose_pack(A, pk, rp)
    assert torch.equal(pk, ops.pack_weight(t))
    sub = A[:, 64:128] if cols >= 128 else A                  # strided source view
    t2 = torch.empty(sub.shape[1], rp, dtype=torch.bfloat16, device=dev)
    T.transpose_pad(sub, t2, rp)
    assert torch.equal(t2.cpu().float()[:, :rows], sub.cpu().float().t())


@pytest.mark.parametrize("Tn,R,N,trans", [(300, 64, 192, False), (4608, 192, 4096, True), (1000, 128, 1088, True), (77, 64, 64, False),
                                          (4608, 64, 22016, False)])
def test_gemm_tn_small(dev, Tn, R, N, trans):
    """C = Pᵀ·Q over the rows (LoRA adapter gradients), with and without the split-T workspace."""
    from bridgelang_amd import train_ops as T
    P_, Q_ = rand_bf16((Tn, R), Tn + R), rand_bf16((Tn, N), N)
    ref = P_.t() @ Q_
    ref = ref.t().contiguous() if trans else ref
    for ws in (None, torch.empty(4 << 20, device=dev)):
        C = torch.full(ref.shape, float("nan"), device=dev)
        T.gemm_tn_small(dv(P_, dev), dv(Q_, dev), C, trans, ws)
        assert torch.allclose(C.cpu(), ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item()), (C.cpu() - ref).abs().max()


def test_pack_into_windows(dev):
    """The packers with a destination window: [W | B] and [Wt | At] assembled from parts equal packing the concatenation."""
    from bridgelang_amd import ops, train_ops as T
    n, k, R = 192, 256, 64
    w, b, a = rand_bf16((n, k), 1), rand_bf16((n, R), 2), rand_bf16((R, k), 3)
    KT, NT = (k + R) // 32, (n + R) // 32
    We = torch.zeros(n // 16, KT, 64, 8, dtype=torch.bfloat16, device=dev)
    T.pack_into(dv(w, dev), We, KT, 0)
    T.pack_into(dv(b, dev), We, KT, k // 32)
    assert torch.equal(We, ops.pack_weight(dv(torch.cat([w, b], 1), dev)))
    WTe = torch.zeros(k // 16, NT, 64, 8, dtype=torch.bfloat16, device=dev)
    T.transpose_pack_into(dv(w, dev), WTe, n, NT, 0)
    T.transpose_pack_into(dv(a, dev), WTe, R, NT, n // 32)
    assert torch.equal(WTe, ops.pack_weight(dv(torch.cat([w.t(), a.t()], 1).contiguous(), dev)))


# the last shape runs on the 256 x 256 tile kernel (576 tiles, persistent walk, ragged last row tile): whole-tile epilogue path
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (9472 // 8, 1024, 512), (70, 64, 192), (4500, 8192, 256)])
def test_training_epilogues_equal_the_two_kernel_forms(dev, M, N, K):
    """The training-step GEMM epilogues (BL_EPI_SWIGLU_KEEP / BIAS_GELU_KEEP / SWIGLU_BWD / GELU_BWD) must reproduce, bit for
    bit, the plain epilogue followed by the elementwise kernel they replace (bl_swiglu_bf16, bl_gelu_bf16,
    bl_swiglu_backward_bf16, bl_gelu_backward_bf16 — each checked against autograd above)."""
    from bridgelang_amd import ops, train_ops as T
    x, w, bias = dv(rand_bf16((M, K), 1), dev), rand_bf16((N, K), 2, 0.1), dv(rand_bf16((N,), 3), dev)
    wp = ops.pack_weight(dv(w, dev))
    bits = lambda t: t.view(torch.int16)
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
    # forward, SwiGLU: rows of W interleave gate / up
    pre_a, act_a, pre_b, act_b = z(M, N), z(M, N // 2), z(M, N), z(M, N // 2)
    ops.gemm(x, wp, pre_a, ops.EPI_NONE); T.swiglu(pre_a, act_a)
    ops.gemm(x, wp, pre_b, ops.EPI_SWIGLU_KEEP, out2=act_b)
    assert torch.equal(bits(pre_a), bits(pre_b)) and torch.equal(bits(act_a), bits(act_b))
    only = z(M, N // 2)
    ops.gemm(x, wp, only, ops.EPI_SWIGLU)                           # the inference epilogue: same activation
    assert torch.equal(bits(only), bits(act_b))
    # forward, bias + GELU
    g_a, g_b, t_b = z(M, N), z(M, N), z(M, N)
    ops.gemm(x, wp, pre_a, ops.EPI_BIAS, bias=bias); T.gelu(pre_a, g_a)
    ops.gemm(x, wp, t_b, ops.EPI_BIAS_GELU_KEEP, bias=bias, out2=g_b)
    assert torch.equal(bits(pre_a), bits(t_b)) and torch.equal(bits(g_a), bits(g_b))
    # backward: dy [M, K'] · Wt → d act [M, N'] with the saved pre-activation
    saved_gu, saved_t = dv(rand_bf16((M, 2 * N), 5, 2.0), dev), dv(rand_bf16((M, N), 6, 2.0), dev)
    d_act, dgu_a, dgu_b = z(M, N), z(M, 2 * N), z(M, 2 * N)
    ops.gemm(x, wp, d_act, ops.EPI_NONE); T.swiglu_backward(saved_gu, d_act, dgu_a)
    ops.gemm(x, wp, dgu_b, ops.EPI_SWIGLU_BWD, res=saved_gu)
    assert torch.equal(bits(dgu_a), bits(dgu_b)) and dgu_b.float().abs().max() > 0
    dt_a, dt_b = z(M, N), z(M, N)
    T.gelu_backward(saved_t, d_act, dt_a)
    ops.gemm(x, wp, dt_b, ops.EPI_GELU_BWD, res=saved_t)
    assert torch.equal(bits(dt_a), bits(dt_b)) and dt_b.float().abs().max() > 0


@pytest.mark.parametrize("rows,cols,p", [(300, 256, 0.1), (77, 1152, 0.5), (64, 64, 0.0)])
def test_lora_dropout_kernels(dev, rows, cols, p):
    """bl_dropout_bf16 / bl_dropout_grad_fix_bf16 against the host restatement of the counter-based mask
    (oracle/synth.py::dropout_keep): same keep pattern, nn.Dropout's 1/(1-p) scaling, and the input-gradient correction
    dy·W + u → dy·W + mask/(1-p) ⊙ u; a different device seed or salt gives a different mask."""
    from bridgelang_amd import train_ops as T
    from oracle.synth import dropout_keep
    x, u, dx0 = rand_bf16((rows, cols), 1), rand_bf16((rows, cols), 2), rand_bf16((rows, cols), 3)
    seed = torch.tensor([41], dtype=torch.int32, device=dev)
    out = torch.full((rows, cols), 9.0, dtype=torch.bfloat16, device=dev)
    T.dropout(dv(x, dev), out, p, seed, 7)
    keep = torch.from_numpy(dropout_keep(41, 7, rows, cols, p))
    inv = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32)
    want = torch.where(keep, P.rb(x * inv), torch.zeros(()))
    assert torch.equal(out.cpu().float(), want)
    if p > 0:
        assert abs(keep.float().mean().item() - (1 - p)) < 0.02
        other = torch.empty_like(out)
        T.dropout(dv(x, dev), other, p, seed, 8)
        assert not torch.equal(other, out)
        seed2 = torch.tensor([42], dtype=torch.int32, device=dev)
        T.dropout(dv(x, dev), other, p, seed2, 7)
        assert not torch.equal(other, out)
    else:
        assert bool(keep.all())
    DX = dv(dx0, dev)
    T.dropout_grad_fix(dv(u, dev), DX, p, seed, 7)
    want_dx = P.rb(dx0 + torch.where(keep, P.rb(u * (inv - 1.0)), -u))
    assert torch.equal(DX.cpu().float(), want_dx)
