"""GPU parity tests, one per C-ABI entry point: HIP kernel vs the CPU oracle (oracle/restate.py, oracle/synth.py) on
the same seeded bf16 inputs. Integer/index results are bit-exact; bf16 results must agree to within one bf16 rounding
(the only legitimate difference is fp32 accumulation order), which the tests express as rtol = 2^-6 on ≥ 99 % exactly
equal elements.
"""
import math

import numpy as np
import pytest
import torch

from conftest import rand_bf16
from oracle import restate as R
from oracle import synth as S

pytestmark = pytest.mark.gpu

P = R.Prec(True)


def dv(x, dev):
    return x.to(torch.bfloat16).to(dev)


def pk(w, dev):
    """nn.Linear-layout CPU weight → packed (fragment-major) device weight."""
    from bridgelang_amd import ops
    return ops.pack_weight(dv(w, dev))


def close_bf16(got, ref, what, rtol=2 ** -6, atol_scale=2 ** -8, min_exact=0.98):
    got = got.float().cpu()
    ref = ref.float()
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs()
    bound = rtol * ref.abs() + atol_scale * scale
    exact = (got == ref).float().mean().item()
    assert (err <= bound).all(), (f"{what}: max err {err.max().item():.4g} (scale {scale:.4g}), "
                                  f"{(err > bound).sum().item()} elements out of tolerance, exact {exact:.4f}")
    assert exact >= min_exact, f"{what}: only {exact:.4f} of elements bit-equal to the oracle"


# ---- synthetic generator: bit exact -------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,mean,std", [((1000,), 0.0, 0.02), ((37, 588), 1.0, 0.02), ((4096,), 0.1, 0.02)])
def test_fill_synth_bit_exact(dev, shape, mean, std):
    from bridgelang_amd import ops
    seed = S.tensor_seed("some.tensor.name", 3)
    n = int(np.prod(shape))
    got = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    ops.fill_synth(got, seed, mean, std / S.IRWIN_HALL_SD)
    ref = S.synth_bf16(shape, seed, mean, std).view(-1)
    assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))


def test_fill_synth_padded_and_interleaved(dev):
    from bridgelang_amd import ops
    rows, cols, ld = 33, 588, 640
    seed = 1234
    got = torch.zeros(rows, ld, dtype=torch.bfloat16, device=dev)
    ops.fill_synth(got.view(-1), seed, 0.0, 0.02 / S.IRWIN_HALL_SD, rows=rows, cols=cols, ld=ld)
    ref = S.synth_bf16((rows, cols), seed, 0.0, 0.02)
    assert torch.equal(got.cpu()[:, :cols], ref)
    assert (got.cpu()[:, cols:] == 0).all()
    # interleave two logical matrices row-wise (the gate/up packing)
    I, D = 48, 64
    gu = torch.zeros(2 * I, D, dtype=torch.bfloat16, device=dev)
    ops.fill_synth(gu.view(-1), 11, 0.0, 0.02 / S.IRWIN_HALL_SD, rows=I, cols=D, ld=2 * D)
    ops.fill_synth(gu.view(-1)[D:], 12, 0.0, 0.02 / S.IRWIN_HALL_SD, rows=I, cols=D, ld=2 * D)
    g, u = S.synth_bf16((I, D), 11, 0.0, 0.02), S.synth_bf16((I, D), 12, 0.0, 0.02)
    assert torch.equal(gu.cpu()[0::2], g) and torch.equal(gu.cpu()[1::2], u)


def test_pack_weight(dev):
    from bridgelang_amd import ops
    w = rand_bf16((4304 // 16 * 16, 1088), 3)
    got = ops.pack_weight(dv(w, dev)).cpu().float()
    n, k = w.shape
    ref = w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(n // 16, k // 32, 64, 8)
    assert torch.equal(got, ref)
    assert torch.equal(ops.unpack_weight(ops.pack_weight(dv(w, dev))).cpu().float(), w)


# ---- tiled GEMM -----------------------------------------------------------------------------------------------------
# the last three are large enough for the 256x256 pipelined kernel (ragged M and N, odd and even K-tile counts)
GEMM_SHAPES = [(300, 192, 128), (128, 128, 64), (261 * 2, 1152, 576), (77, 4304 // 16 * 16, 256), (1, 64, 64),
               (1044, 528, 320), (2304, 1024, 512), (1100, 4304 // 16 * 16, 1152)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_none_bias_gelu(dev, M, N, K):
    from bridgelang_amd import ops
    a, w, b = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1)
    A, W, Bv = dv(a, dev), pk(w, dev), dv(b, dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, out, ops.EPI_NONE, skinny=False)
    close_bf16(out, R.linear(P, a, w), "EPI_NONE")
    ops.gemm(A, W, out, ops.EPI_BIAS, bias=Bv, skinny=False)
    close_bf16(out, R.linear(P, a, w, b), "EPI_BIAS")
    ops.gemm(A, W, out, ops.EPI_BIAS_GELU, bias=Bv, skinny=False)
    close_bf16(out, R.gelu(P, R.linear(P, a, w, b)), "EPI_BIAS_GELU")
    outf = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(A, W, outf, ops.EPI_F32, skinny=False)
    ref = a @ w.t()
    assert torch.allclose(outf.cpu(), ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item())
    ops.gemm(A, W, outf, ops.EPI_F32_BF16R, skinny=False)
    close_bf16(outf, R.linear(P, a, w), "EPI_F32_BF16R")


@pytest.mark.parametrize("M,N,K,epi", [(4608, 4096, 512, "res"), (4608, 1024, 1024, "gelu"), (4400, 3584, 768, "swiglu"),
                                       (4608, 12288, 256, "none"),
                                       # few-tile (tall-skinny) problems: the 128 kernel slices K over grid.y
                                       (4608, 64, 4096, "none"), (4176, 192, 3072, "res"), (300, 128, 2048, "gelu"),
                                       (1000, 256, 1536, "swiglu")])
def test_gemm_splitk_tail(dev, M, N, K, epi):
    """Shapes whose 256x256 tile count is not a multiple of 256: whole rounds + split-K tail (with a workspace) must
    equal the oracle, and equal the no-workspace dispatch bit for bit except for fp32 summation order."""
    from bridgelang_amd import ops
    a, w, b, r = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1), rand_bf16((M, N), 4)
    A, Bv, Rr = dv(a, dev), dv(b, dev), dv(r, dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    if epi == "swiglu":
        I = N // 2
        W = pk(torch.stack([w[:I], w[I:]], 1).reshape(N, K), dev)
        out, out2 = (torch.zeros(M, I, dtype=torch.bfloat16, device=dev) for _ in range(2))
        ops.gemm(A, W, out, ops.EPI_SWIGLU, workspace=ws)
        ops.gemm(A, W, out2, ops.EPI_SWIGLU)
        g, u = R.linear(P, a, w[:I]), R.linear(P, a, w[I:])
        ref = P.rb(P.rb(torch.nn.functional.silu(g)) * u)
    else:
        W = pk(w, dev)
        out, out2 = (torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2))
        code = {"res": ops.EPI_RES, "gelu": ops.EPI_BIAS_GELU, "none": ops.EPI_NONE}[epi]
        kw = {"res": dict(res=Rr), "gelu": dict(bias=Bv), "none": {}}[epi]
        ops.gemm(A, W, out, code, workspace=ws, **kw)
        ops.gemm(A, W, out2, code, **kw)
        lin = R.linear(P, a, w, b if epi == "gelu" else None)
        ref = {"res": lambda: P.rb(r + lin), "gelu": lambda: R.gelu(P, lin), "none": lambda: lin}[epi]()
    close_bf16(out, ref, f"split-K tail {epi}")
    close_bf16(out, out2.cpu().float(), "split-K vs 128-tail dispatch", min_exact=0.97)


@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 2048), (12288, 512)])   # 12288: 96 leftover tiles = a partial fourth round
def test_gemm_rows_independent_of_batch(dev, N, K):
    """Batch invariance at op level: a row's output must be bit-identical whether it is computed alone (M = 288 → the
    128x128 kernel) or inside a 16x larger problem (the 256x256 pipelined kernel, whole rounds + 128-tile tail), for
    every position in the big problem. (No workspace → no split-K, which is the one path that would break this.)"""
    from bridgelang_amd import ops
    S, B = 288, 16
    a, w = rand_bf16((B * S, K), 1), rand_bf16((N, K), 2, 0.05)
    A, W = dv(a, dev), pk(w, dev)
    big = torch.empty(B * S, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, big, ops.EPI_NONE)
    for b in (0, 7, 14, 15):
        one = torch.empty(S, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(A[b * S:(b + 1) * S], W, one, ops.EPI_NONE)
        assert torch.equal(one, big[b * S:(b + 1) * S]), f"sequence {b}"


@pytest.mark.parametrize("M,N,K,epi", [(4608, 4096, 1024, "res"), (4591, 4000, 512, "none"), (4608, 12288, 256, "bias"),
                                       (4608, 2048, 384, "swiglu")])
def test_gemm288_sequence_tiles(dev, M, N, K, epi):
    """288-row tiles (one 288-token sequence per row tile: gemm288_kernel) where they remove the leftover round of 256-row
    tiling — Llama o_proj / down_proj / qkv at 16 × 288 rows — incl. ragged M and N and every epilogue family, against the
    oracle on sampled rows and bit-identical to the 256-row tiling (BL_GEMM_NO_288 is read once per process, so the
    comparison is against the per-sequence call, which takes the mid kernels)."""
    from bridgelang_amd import ops
    a, w, b, r = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1), rand_bf16((M, N), 4)
    A, Bv, Rr = dv(a, dev), dv(b, dev), dv(r, dev)
    sel = torch.cat([torch.arange(0, 200), torch.arange(2200, 2400), torch.arange(M - 200, M)])
    if epi == "swiglu":
        I = N // 2
        W = pk(torch.stack([w[:I], w[I:]], 1).reshape(N, K), dev)
        out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
        kw, e = {}, ops.EPI_SWIGLU
        g, u = R.linear(P, a[sel], w[:I]), R.linear(P, a[sel], w[I:])
        ref = P.rb(P.rb(torch.nn.functional.silu(g)) * u)
    else:
        W = pk(w, dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw, e = {"res": dict(res=Rr), "bias": dict(bias=Bv), "none": {}}[epi], {"res": ops.EPI_RES, "bias": ops.EPI_BIAS, "none": ops.EPI_NONE}[epi]
        ref = {"res": lambda: P.rb(r[sel] + R.linear(P, a[sel], w)), "bias": lambda: R.linear(P, a[sel], w, b),
               "none": lambda: R.linear(P, a[sel], w)}[epi]()
    ops.gemm(A, W, out, e, **kw)
    close_bf16(out[sel], ref, f"gemm288 {epi}")
    for s0 in (0, 288 * 7, M - 288 - (M % 288 and 5)):              # one 288-row sequence at a time: the mid kernels
        one = torch.empty(288, out.shape[1], dtype=torch.bfloat16, device=dev)
        kw1 = {k: (v[s0:s0 + 288] if k == "res" else v) for k, v in kw.items()}
        ops.gemm(A[s0:s0 + 288], W, one, e, **kw1)
        assert torch.equal(one, out[s0:s0 + 288]), f"rows {s0}..: result depends on the tiling"


@pytest.mark.parametrize("M,N,K,epi", [(2333, 15360, 256, "bias"), (2560, 15344, 384, "none"), (2333, 15360, 256, "swiglu"),
                                       (2500, 15360, 128, "res")])
def test_gemm256_persistent_multi_round(dev, M, N, K, epi):
    """More than one round of 256 x 256 tiles with a partial last round of > 64 tiles (600 tiles: 88 workgroups walk three
    tiles, 168 walk two) goes to the PERSISTENT form of gemm256s_kernel — one workgroup per CU, the next tile's first
    K-tiles staged behind this tile's epilogue. Ragged M and N (edge tiles take the per-store epilogue, interior tiles
    the straight-line one), even K-tile counts 2 … 6, four epilogue families: against the oracle on sampled rows, and —
    since a row's K order is the same in every tile kernel — bit-identical to 288-row slices run on their own (mid
    kernels)."""
    from bridgelang_amd import ops
    a, w, b, r = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1), rand_bf16((M, N), 4)
    A, Bv, Rr = dv(a, dev), dv(b, dev), dv(r, dev)
    sel = torch.cat([torch.arange(0, 150), torch.arange(1200, 1350), torch.arange(M - 150, M)])
    if epi == "swiglu":
        I = N // 2
        W = pk(torch.stack([w[:I], w[I:]], 1).reshape(N, K), dev)
        out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
        kw, e = {}, ops.EPI_SWIGLU
        g, u = R.linear(P, a[sel], w[:I]), R.linear(P, a[sel], w[I:])
        ref = P.rb(P.rb(torch.nn.functional.silu(g)) * u)
    else:
        W = pk(w, dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw, e = {"res": dict(res=Rr), "bias": dict(bias=Bv), "none": {}}[epi], {"res": ops.EPI_RES, "bias": ops.EPI_BIAS, "none": ops.EPI_NONE}[epi]
        ref = {"res": lambda: P.rb(r[sel] + R.linear(P, a[sel], w)), "bias": lambda: R.linear(P, a[sel], w, b),
               "none": lambda: R.linear(P, a[sel], w)}[epi]()
    out.fill_(float("nan"))
    ops.gemm(A, W, out, e, **kw)
    assert not torch.isnan(out.float()).any(), "a tile of the persistent walk was never written"
    close_bf16(out[sel], ref, f"gemm256 persistent {epi}")
    for s0 in (0, 288 * 3, M - 288):
        one = torch.empty(288, out.shape[1], dtype=torch.bfloat16, device=dev)
        kw1 = {k: (v[s0:s0 + 288] if k == "res" else v) for k, v in kw.items()}
        ops.gemm(A[s0:s0 + 288], W, one, e, **kw1)
        assert torch.equal(one, out[s0:s0 + 288]), f"rows {s0}..: result depends on the tiling"


@pytest.mark.parametrize("T,N,K", [(261, 1024, 1024), (256, 1152, 4352), (256, 1024, 640), (261, 1024, 1088)])   # last: odd K-tile count (LoRA: K + rank columns)
def test_gemm_ring160_vit_shapes_and_batch_invariance(dev, T, N, K):
    """The narrow ViT layers at 16 images (attn.proj / mlp.fc2 / patch embed: M = 16·T rows, N ≤ 1152) run as ONE round of
    160 × 128 tiles on the ring-buffered kernel (gemm_tail_kernel stand-alone mode): vs the oracle with the fused
    bias + LayerScale + residual epilogue, and bit-identical per image to the one-image call (mid kernels)."""
    from bridgelang_amd import ops
    Bn = 16
    M = Bn * T
    a, w, b = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1)
    r, ls = rand_bf16((M, N), 4), rand_bf16((N,), 5, 0.3)
    A, Bv, Rr, Ls, W = dv(a, dev), dv(b, dev), dv(r, dev), dv(ls, dev), pk(w, dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, out, ops.EPI_BIAS_RES, bias=Bv, scale=Ls, res=Rr)
    sel = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])          # first / last rows (incl. the ragged last tile)
    close_bf16(out[sel], P.rb(r[sel] + P.rb(R.linear(P, a[sel], w, b) * ls)), "ring160 BIAS_RES+LayerScale")
    for img in (0, 9, 15):
        one = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(A[img * T:(img + 1) * T], W, one, ops.EPI_BIAS_RES, bias=Bv, scale=Ls, res=Rr[img * T:(img + 1) * T])
        assert torch.equal(one, out[img * T:(img + 1) * T]), f"image {img}: result depends on the batch it ran in"


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (522, 256, 1088), (1305, 768, 448)])
def test_gemm_residual_layerscale_swiglu(dev, M, N, K):
    from bridgelang_amd import ops
    a, w, b = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1)
    r, ls = rand_bf16((M, N), 4), rand_bf16((N,), 5, 0.3)
    A, Bv, Rr, Ls = (dv(t, dev) for t in (a, b, r, ls))
    W = pk(w, dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, out, ops.EPI_RES, res=Rr, skinny=False)
    close_bf16(out, P.rb(r + R.linear(P, a, w)), "EPI_RES")
    ops.gemm(A, W, out, ops.EPI_BIAS_RES, bias=Bv, res=Rr, skinny=False)
    close_bf16(out, P.rb(r + R.linear(P, a, w, b)), "EPI_BIAS_RES")
    ops.gemm(A, W, out, ops.EPI_BIAS_RES, bias=Bv, scale=Ls, res=Rr, skinny=False)
    close_bf16(out, P.rb(r + P.rb(R.linear(P, a, w, b) * ls)), "EPI_BIAS_RES+LayerScale")
    # in-place residual (out aliases res), as the engine uses it
    x = Rr.clone()
    ops.gemm(A, W, x, ops.EPI_RES, res=x, skinny=False)
    close_bf16(x, P.rb(r + R.linear(P, a, w)), "EPI_RES in place")
    # SwiGLU with interleaved gate/up rows
    I = N // 2
    gate, up = w[:I], w[I:]
    gu = torch.stack([gate, up], 1).reshape(N, K)
    out2 = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, pk(gu, dev), out2, ops.EPI_SWIGLU, skinny=False)
    g, u = R.linear(P, a, gate), R.linear(P, a, up)
    close_bf16(out2, P.rb(P.rb(torch.nn.functional.silu(g)) * u), "EPI_SWIGLU")


def test_gemm_row_remap_and_table_residual(dev):
    """patch-embed style: + pos[m % 256], rows written behind 5 prefix tokens; and the tap: drop 5 prefix rows."""
    from bridgelang_amd import ops
    B, D, K = 2, 128, 640
    a, w, b, pos = rand_bf16((B * 256, K), 1), rand_bf16((D, K), 2, 0.05), rand_bf16((D,), 3, 0.1), rand_bf16((256, D), 4)
    out = torch.full((B * 261, D), 7.0, dtype=torch.bfloat16, device=dev)
    ops.gemm(dv(a, dev), pk(w, dev), out, ops.EPI_BIAS_RES, bias=dv(b, dev), res=dv(pos, dev), res_row_mod=256,
             out_map=(256, 261, 5))
    ref = P.rb(R.linear(P, a, w, b).view(B, 256, D) + pos)
    got = out.cpu().float().view(B, 261, D)
    close_bf16(got[:, 5:], ref, "patch-embed remap")
    assert (got[:, :5] == 7.0).all(), "prefix rows must not be touched"
    # tap: input rows [B*261], output only the 256 patch rows per image, into a wider concat buffer
    x = rand_bf16((B * 261, K), 9)
    cat = torch.zeros(B * 256, D + 64, dtype=torch.bfloat16, device=dev)
    ops.gemm(dv(x, dev), pk(w, dev), cat[:, 64:], ops.EPI_BIAS, bias=dv(b, dev), out_map=(261, 256, -5))
    ref2 = R.linear(P, x, w, b).view(B, 261, D)[:, 5:].reshape(B * 256, D)
    close_bf16(cat[:, 64:], ref2, "tap remap")
    assert (cat[:, :64] == 0).all()


# ---- skinny GEMM ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(16, 512, 512), (1, 1536 * 2, 512), (16, 512, 1536), (7, 4096, 4096), (16, 256, 11008)])
def test_gemm_skinny(dev, M, N, K):
    from bridgelang_amd import ops
    a, w, r = rand_bf16((M, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((M, N), 4)
    A, W, Rr = dv(a, dev), pk(w, dev), dv(r, dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, out, ops.EPI_NONE, skinny=True)
    close_bf16(out, R.linear(P, a, w), "skinny NONE")
    ops.gemm(A, W, out, ops.EPI_RES, res=Rr, skinny=True)
    close_bf16(out, P.rb(r + R.linear(P, a, w)), "skinny RES")
    I = N // 2
    gu = torch.stack([w[:I], w[I:]], 1).reshape(N, K)
    out2 = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, pk(gu, dev), out2, ops.EPI_SWIGLU, skinny=True)
    g, u = R.linear(P, a, w[:I]), R.linear(P, a, w[I:])
    close_bf16(out2, P.rb(P.rb(torch.nn.functional.silu(g)) * u), "skinny SWIGLU")
    outf = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(A, W, outf, ops.EPI_F32_BF16R, skinny=True)
    close_bf16(outf, R.linear(P, a, w), "skinny F32_BF16R")


# ---- norms ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,dim", [(9, 1024), (261, 1152), (5, 576), (33, 4096), (3, 5120), (2, 256)])
def test_norms(dev, rows, dim):
    from bridgelang_amd import ops
    x, w, b = rand_bf16((rows, dim), 1, 3.0), rand_bf16((dim,), 2, 0.02) + 1, rand_bf16((dim,), 3, 0.02)
    w = P.rb(w)
    X, W, Bv = dv(x, dev), dv(w, dev), dv(b, dev)
    y = torch.empty_like(X)
    ops.layernorm(X, W, Bv, y, 1e-6)
    close_bf16(y, R.layernorm(P, x, w, b, 1e-6), "layernorm")
    ops.rmsnorm(X, W, y, 1e-6)
    close_bf16(y, R.rmsnorm(P, x, w, 1e-6), "rmsnorm")


# ---- attention ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hd,H,Sq,causal", [(64, 4, 261, False), (72, 8, 256, False), (128, 4, 288, True),
                                            (128, 2, 70, True), (64, 2, 17, False), (128, 2, 296, True)])
def test_attention_prefill(dev, hd, H, Sq, causal):
    from bridgelang_amd import ops
    B, D = 2, H * hd
    qkv = rand_bf16((B * Sq, 3 * D), hd + Sq)
    Q = dv(qkv, dev)
    o = torch.zeros(B * Sq, D, dtype=torch.bfloat16, device=dev)
    st = (Sq * 3 * D, hd, 3 * D)
    ops.attention(Q, Q[:, D:], Q[:, 2 * D:], o, B=B, H=H, Sq=Sq, Skv=Sq, head_dim=hd, q_strides=st, k_strides=st,
                  v_strides=st, o_strides=(Sq * D, hd, D), causal=causal)
    t = qkv.view(B, Sq, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = R.attention(P, t[0], t[1], t[2], hd ** -0.5, causal).permute(0, 2, 1, 3).reshape(B * Sq, D)
    close_bf16(o, ref, f"attention hd={hd}", rtol=2 ** -5, atol_scale=2 ** -7, min_exact=0.98)


@pytest.mark.parametrize("hd,H,Sq,causal", [(128, 2, 336, True), (128, 2, 406, True), (64, 2, 700, False), (72, 2, 333, False)])
def test_attention_long_sequences_chunked_kernel(dev, hd, H, Sq, causal):
    """Sequences beyond 320 positions (prompts longer than 64 tokens) take the chunked online-softmax kernel
    (attn_fwd_kernel): same oracle, bf16-rounding tolerance — online rescaling changes the fp32 rounding of the
    probabilities, so fewer outputs are bit-equal than with the exact two-pass kernel of the short path."""
    from bridgelang_amd import ops
    B, D = 2, H * hd
    qkv = rand_bf16((B * Sq, 3 * D), hd + Sq)
    Q = dv(qkv, dev)
    o = torch.zeros(B * Sq, D, dtype=torch.bfloat16, device=dev)
    st = (Sq * 3 * D, hd, 3 * D)
    ops.attention(Q, Q[:, D:], Q[:, 2 * D:], o, B=B, H=H, Sq=Sq, Skv=Sq, head_dim=hd, q_strides=st, k_strides=st,
                  v_strides=st, o_strides=(Sq * D, hd, D), causal=causal)
    t = qkv.view(B, Sq, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = R.attention(P, t[0], t[1], t[2], hd ** -0.5, causal).permute(0, 2, 1, 3).reshape(B * Sq, D)
    close_bf16(o, ref, f"chunked attention hd={hd} S={Sq}", rtol=2 ** -5, atol_scale=2 ** -7, min_exact=0.5)


def test_attention_key_mask(dev):
    from bridgelang_amd import ops
    B, H, hd, S = 3, 2, 128, 100
    D = H * hd
    qkv = rand_bf16((B * S, 3 * D), 5)
    lens = [100, 63, 17]
    mask = torch.zeros(B, S, dtype=torch.uint8)
    for i, n in enumerate(lens):
        mask[i, :n] = 1
    Q = dv(qkv, dev)
    o = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
    st = (S * 3 * D, hd, 3 * D)
    ops.attention(Q, Q[:, D:], Q[:, 2 * D:], o, B=B, H=H, Sq=S, Skv=S, head_dim=hd, q_strides=st, k_strides=st,
                  v_strides=st, o_strides=(S * D, hd, D), causal=True, key_mask=mask.to(dev))
    t = qkv.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = R.attention(P, t[0], t[1], t[2], hd ** -0.5, True, key_mask=mask).permute(0, 2, 1, 3)
    got = o.cpu().float().view(B, S, H, hd)
    for i, n in enumerate(lens):     # rows of padded queries are unspecified (ignored by the loss)
        close_bf16(got[i, :n], ref[i, :n], f"masked attention b={i}", rtol=2 ** -5, atol_scale=2 ** -7, min_exact=0.98)


@pytest.mark.parametrize("Skv", [1, 67, 289, 294])
def test_attention_decode(dev, Skv):
    from bridgelang_amd import ops
    B, H, hd, cache_len = 3, 4, 128, 320
    D = H * hd
    q = rand_bf16((B, D), 1)
    kc, vc = rand_bf16((B, H, cache_len, hd), 2), rand_bf16((B, H, cache_len, hd), 3)
    o = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
    cs = (H * cache_len * hd, cache_len * hd, hd)
    ops.attention_decode(dv(q, dev), dv(kc, dev), dv(vc, dev), o, B=B, H=H, Skv=Skv, head_dim=hd,
                         q_strides=(D, hd, D), k_strides=cs, v_strides=cs, o_strides=(D, hd, D))
    ref = R.attention(P, q.view(B, H, 1, hd), kc[:, :, :Skv], vc[:, :, :Skv], hd ** -0.5, False).reshape(B, D)
    close_bf16(o, ref, "decode attention", rtol=2 ** -5, atol_scale=2 ** -7, min_exact=0.98)


def test_attention_decode_rope_grouped(dev):
    """One grouped launch ≡ one bl_attention_decode_rope_bf16 launch per group, bit for bit (outputs and cache rows)."""
    from bridgelang_amd import ops
    B, H, hd, cache_len, G = 3, 4, 128, 320, 6
    D = H * hd
    cos, sin = R.rope_tables(hd, 512, 10000.0)
    C_, S_ = dv(cos, dev), dv(sin, dev)
    qkv = dv(rand_bf16((G * B, 3 * D), 21), dev)
    pos = [288 + g for g in range(G)]
    kc = [dv(rand_bf16((B, H, cache_len, hd), 30 + g), dev) for g in range(G)]
    vc = [dv(rand_bf16((B, H, cache_len, hd), 40 + g), dev) for g in range(G)]
    kc2, vc2 = [t.clone() for t in kc], [t.clone() for t in vc]
    o1 = torch.zeros(G * B, D, dtype=torch.bfloat16, device=dev)
    o2 = torch.zeros_like(o1)
    for g in range(G):
        r = slice(g * B, (g + 1) * B)
        ops.attention_decode_rope(qkv[r], kc[g], vc[g], o1[r], C_, S_, B=B, H=H, head_dim=hd, pos=pos[g])
    ops.attention_decode_rope_grouped(qkv, kc2, vc2, o2, C_, S_, B=B, H=H, head_dim=hd, pos=pos)
    assert torch.equal(o1, o2)
    for g in range(G):
        assert torch.equal(kc[g], kc2[g]) and torch.equal(vc[g], vc2[g])
    with pytest.raises(ValueError):
        ops.attention_decode_rope_grouped(qkv, kc2, vc2, o2, C_, S_, B=B, H=H, head_dim=hd, pos=[320] * G)


# ---- Llama glue -----------------------------------------------------------------------------------------------------
def test_rope_kvcache(dev):
    from bridgelang_amd import ops
    B, S, H, hd, cache_len, pos0 = 2, 37, 4, 128, 64, 3
    D = H * hd
    qkv = rand_bf16((B * S, 3 * D), 8)
    cos, sin = R.rope_tables(hd, 256, 10000.0)
    Q = dv(qkv, dev)
    kc = torch.zeros(B, H, cache_len, hd, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    ops.rope_kvcache(Q, dv(cos, dev), dv(sin, dev), kc, vc, B=B, S=S, H=H, head_dim=hd, pos0=pos0)
    t = qkv.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)       # [3, B, H, S, hd]
    q_ref, k_ref = R.apply_rope(P, t[0], cos, sin, pos0), R.apply_rope(P, t[1], cos, sin, pos0)
    got = Q.cpu().float().view(B, S, 3, H, hd)
    assert torch.equal(got[:, :, 0].permute(0, 2, 1, 3), q_ref), "rotated q must be bit-exact"
    assert torch.equal(got[:, :, 2], qkv.view(B, S, 3, H, hd)[:, :, 2]), "v must be untouched"
    assert torch.equal(kc.cpu().float()[:, :, pos0:pos0 + S], k_ref), "cached k must be bit-exact"
    assert torch.equal(vc.cpu().float()[:, :, pos0:pos0 + S], t[2]), "cached v must be bit-exact"
    assert (kc.cpu()[:, :, :pos0] == 0).all() and (kc.cpu()[:, :, pos0 + S:] == 0).all()


@pytest.mark.parametrize("B,H,S,pos0,masked", [(2, 4, 37, 3, False), (2, 32, 288, 0, False), (4, 32, 70, 0, True), (3, 8, 296, 5, True),
                                                (1, 2, 320, 0, False)])
def test_attention_rope_fused(dev, B, H, S, pos0, masked):
    """bl_attention_rope_bf16 == bl_rope_kvcache_bf16 followed by bl_attention_bf16, bit for bit (outputs and caches), and
    its caches are bit-exact against the oracle's apply_rope."""
    from bridgelang_amd import ops
    hd, cache_len = 128, 336
    D = H * hd
    qkv = rand_bf16((B * S, 3 * D), 11 + S)
    cos, sin = R.rope_tables(hd, 512, 10000.0)
    C_, S_ = dv(cos, dev), dv(sin, dev)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.uint8)
        for i in range(B):
            mask[i, S - 3 * i:] = 0
        mask = mask.to(dev)
    # two-kernel path
    Q1 = dv(qkv, dev)
    kc1 = torch.zeros(B, H, cache_len, hd, dtype=torch.bfloat16, device=dev)
    vc1 = torch.zeros_like(kc1)
    o1 = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
    ops.rope_kvcache(Q1, C_, S_, kc1, vc1, B=B, S=S, H=H, head_dim=hd, pos0=pos0)
    cs = (H * cache_len * hd, cache_len * hd, hd)
    ops.attention(Q1, kc1[:, :, pos0:], vc1[:, :, pos0:], o1, B=B, H=H, Sq=S, Skv=S, head_dim=hd, q_strides=(S * 3 * D, hd, 3 * D),
                  k_strides=cs, v_strides=cs, o_strides=(S * D, hd, D), causal=True, key_mask=mask)
    # fused
    Q2 = dv(qkv, dev)
    kc2 = torch.zeros_like(kc1)
    vc2 = torch.zeros_like(kc1)
    o2 = torch.zeros_like(o1)
    ops.attention_rope(Q2, kc2, vc2, o2, C_, S_, B=B, S=S, H=H, head_dim=hd, pos0=pos0, key_mask=mask)
    assert torch.equal(Q2.cpu(), qkv.to(torch.bfloat16)), "the fused kernel must leave qkv untouched"
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2), "caches differ from bl_rope_kvcache_bf16"
    t = qkv.view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    assert torch.equal(kc2.cpu().float()[:, :, pos0:pos0 + S], R.apply_rope(P, t[1], cos, sin, pos0)), "cached k vs oracle"
    assert torch.equal(vc2.cpu().float()[:, :, pos0:pos0 + S], t[2])
    assert (kc2.cpu()[:, :, :pos0] == 0).all() and (kc2.cpu()[:, :, pos0 + S:] == 0).all()
    a, b = o1.view(B, S, D), o2.view(B, S, D)
    if masked:       # rows of masked-out queries are unspecified
        for i in range(B):
            assert torch.equal(a[i, :S - 3 * i], b[i, :S - 3 * i]), f"fused attention differs, b={i}"
    else:
        assert torch.equal(a, b), "fused attention differs from the two-kernel path"


def test_attention_rope_rejects(dev):
    from bridgelang_amd import ops, _lib
    B, H, hd, S = 1, 2, 128, 32
    D = H * hd
    qkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device=dev)
    kc = torch.zeros(B, H, 40, hd, dtype=torch.bfloat16, device=dev)
    o = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
    cos = torch.zeros(64, 64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.BridgeLangHipError):       # pos0 + S beyond the cache
        ops.attention_rope(qkv, kc, kc.clone(), o, cos, cos, B=B, S=S, H=H, head_dim=hd, pos0=9)


def test_embed_splice_argmax_im2col_prefix(dev):
    from bridgelang_amd import ops
    B, L, dim, V, npatch = 3, 9, 64, 100, 16
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (B, L), generator=g)
    table = rand_bf16((V, dim), 1)
    dst = torch.full((B, L + npatch, dim), 5.0, dtype=torch.bfloat16, device=dev)
    ops.embed_splice(ids.to(dev), dv(table, dev), dst, npatch)
    got = dst.cpu().float()
    assert torch.equal(got[:, 0], table[ids[:, 0]]) and torch.equal(got[:, 1 + npatch:], table[ids[:, 1:]])
    assert (got[:, 1:1 + npatch] == 5.0).all()
    # argmax with ties → first index
    lg = torch.randn(5, 32064, generator=g)
    lg[1, 100] = lg[1, 31000] = 50.0
    lg[2, 32063] = 60.0
    lg[3, 0] = 70.0
    out = torch.zeros(5, dtype=torch.int64, device=dev)
    ops.argmax(lg.to(dev), out)
    assert torch.equal(out.cpu(), lg.argmax(-1)) and out[1].item() == 100
    # im2col
    pv = rand_bf16((2, 6, 224, 224), 3)
    for chan0 in (0, 3):
        col = torch.full((2 * 256, 640), 9.0, dtype=torch.bfloat16, device=dev)
        ops.im2col_patch14(dv(pv, dev), chan0, col)
        ref = R.im2col_patch14(pv[:, chan0:chan0 + 3].contiguous()).reshape(2 * 256, 588)
        assert torch.equal(col.cpu().float()[:, :588], ref) and (col.cpu()[:, 588:] == 0).all()
    # prefix tokens
    x = torch.zeros(B * 21, dim, dtype=torch.bfloat16, device=dev)
    pre = rand_bf16((5, dim), 4)
    ops.write_prefix_tokens(dv(pre, dev), x, B, 21)
    got = x.cpu().float().view(B, 21, dim)
    assert torch.equal(got[:, :5], pre.expand(B, 5, dim)) and (got[:, 5:] == 0).all()


def test_preprocess_uint8_frames_bit_exact(dev):
    """GPU to_tensor + dual normalise + channel stack of frames at the model resolution ≡ the host image processor."""
    from PIL import Image
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, (3, 224, 224, 3), dtype=np.uint8)
    frames[0, :2] = 0
    frames[1, :2] = 255
    ip = PrismaticImageProcessor()
    ref = torch.stack([ip.apply_transform(Image.fromarray(f)) for f in frames]).to(torch.bfloat16)
    got = ip.preprocess_frames_gpu(torch.from_numpy(frames).to(dev))
    assert got.shape == (3, 6, 224, 224) and torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("H,W", [(256, 256), (480, 640), (100, 300), (224, 500), (300, 224), (128, 96), (720, 1280)])
def test_resize_bicubic_frames_bit_exact(dev, H, W):
    """bl_resample_pass_u8 (horizontal + vertical 8-bit passes) ≡ Pillow's bicubic `Image.resize` bit for bit, and the
    whole GPU frame path ≡ the host image processor (resize → to_tensor → dual normalise → stack → bf16)."""
    from PIL import Image
    from bridgelang_amd import ops
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor
    from oracle import resample as RS
    rng = np.random.default_rng(H * 10007 + W)
    frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    frames[1, : H // 2] = 255                      # hard edges: over/undershoot must clip exactly as clip8 does
    frames[1, H // 2:] = 0
    frames[2, :, ::2] = 0
    F = torch.from_numpy(frames).to(dev)
    got = ops.resize_bicubic_u8(F, 224, 224)
    assert torch.equal(got.cpu(), torch.from_numpy(RS.pil_resize(frames, 224, 224)))
    ip = PrismaticImageProcessor()
    ref = torch.stack([ip.apply_transform(Image.fromarray(f)) for f in frames]).to(torch.bfloat16)
    assert torch.equal(ip.preprocess_frames_gpu(F).cpu(), ref)


@pytest.mark.parametrize("H,W,out_hw,crop_scale", [(256, 256, (224, 224), 0.9), (480, 640, (224, 224), 0.9), (224, 224, (224, 224), 1.0),
                                                   (128, 96, (224, 224), 0.5), (720, 1280, (224, 224), 0.9), (100, 77, (13, 1), 0.5),
                                                   (64, 64, (224, 224), 1.3)])
def test_center_crop_and_resize_device_twin_bit_exact(dev, H, W, out_hw, crop_scale):
    """bl_crop_resize_bilinear_u8 (eval-time centre crop of the robot loops, experiments/robot/openvla_utils.py:81-155)
    ≡ the host restatement of tf.image.crop_and_resize + convert_image_dtype, bit for bit, on uint8 frames in HBM
    (TensorFlow itself is absent: the restatement is unpinned against TF, the kernel is pinned against the restatement)."""
    from bridgelang_amd.vla import eval_preprocess as EP
    rng = np.random.default_rng(H * 131 + W)
    frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    frames[1, : H // 2] = 255
    frames[1, H // 2:] = 0
    frames[2, :, ::2] = 0
    want = np.stack([EP.center_crop_and_resize(f, crop_scale, out_hw) for f in frames])
    got = EP.center_crop_and_resize_gpu(torch.from_numpy(frames).to(dev), crop_scale, out_hw)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (3, out_hw[0], out_hw[1], 3)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("H,W,size", [(256, 256, (224, 224)), (480, 640, (224, 224)), (128, 128, (224, 224)), (224, 300, (224, 224)), (256, 256, (128, 160))])
def test_lanczos_resize_device_twin_bit_exact(dev, H, W, size):
    """The resize half of LIBERO's resize_image (libero_utils.py:33-47) on the device: bl_resample_pass_u8 with Lanczos-3
    tables ≡ Pillow's `Image.resize(LANCZOS)` — the library the host restatement calls — bit for bit; with the JPEG
    round trip done on the host the whole function matches."""
    from PIL import Image
    from bridgelang_amd.vla import eval_preprocess as EP
    rng = np.random.default_rng(H * 17 + W)
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    frames[1, : H // 2] = 255
    frames[1, H // 2:] = 0
    h, w = size
    want = np.stack([np.asarray(Image.fromarray(f).resize((w, h), Image.LANCZOS), dtype=np.uint8) for f in frames])
    got = EP.lanczos_resize_gpu(torch.from_numpy(frames).to(dev), size)
    assert np.array_equal(got.cpu().numpy(), want)
    jp = np.stack([EP.jpeg_round_trip(f) for f in frames])
    assert np.array_equal(EP.lanczos_resize_gpu(torch.from_numpy(jp).to(dev), size).cpu().numpy(),
                          np.stack([EP.resize_image(f, size) for f in frames]))


@pytest.mark.parametrize("M,N,K,epi", [(288, 12288, 4096, "none"), (288, 4096, 1024, "res"), (261, 1024, 4096, "res"), (256, 4352, 1152, "gelu"),
                                       (100, 192, 512, "gelu"), (33, 64, 1536, "none"), (320, 3072, 1536, "swiglu"), (288, 256, 1088, "res"),
                                       (576, 4096, 1024, "res"), (640, 12288, 512, "none"), (522, 1024, 1024, "gelu"), (161, 32064, 512, "none"),
                                       (600, 22016, 512, "swiglu")])
def test_gemm_mid_rows(dev, M, N, K, epi):
    """16 < M <= 640 (batch-1 / batch-2 prefill, one or two images' tokens): the weight-streaming mid kernels (mid: all
    rows per column slab; mid2: 160-row workgroups, 32 / 64 / 128 columns) vs the oracle, bit-identical to the tile kernels
    (same K order) so a row's result does not depend on how many rows ran with it; K-sliced with a workspace stays
    within rounding."""
    from bridgelang_amd import ops
    Mbig = 700
    a, w, b, r = rand_bf16((Mbig, K), 1), rand_bf16((N, K), 2, 0.05), rand_bf16((N,), 3, 0.1), rand_bf16((Mbig, N), 4)
    A, Bv, Rr = dv(a, dev), dv(b, dev), dv(r, dev)
    if epi == "swiglu":
        I = N // 2
        W = pk(torch.stack([w[:I], w[I:]], 1).reshape(N, K), dev)
        code, kw, ncols = ops.EPI_SWIGLU, {}, I
        g, u = R.linear(P, a[:M], w[:I]), R.linear(P, a[:M], w[I:])
        ref = P.rb(P.rb(torch.nn.functional.silu(g)) * u)
    else:
        W = pk(w, dev)
        code = {"res": ops.EPI_RES, "gelu": ops.EPI_BIAS_GELU, "none": ops.EPI_NONE}[epi]
        kw, ncols = {"res": dict(res=Rr), "gelu": dict(bias=Bv), "none": {}}[epi], N
        lin = R.linear(P, a[:M], w, b if epi == "gelu" else None)
        ref = {"res": lambda: P.rb(r[:M] + lin), "gelu": lambda: R.gelu(P, lin), "none": lambda: lin}[epi]()
    out, big, out_ws = (torch.zeros(Mbig, ncols, dtype=torch.bfloat16, device=dev) for _ in range(3))
    kws = {k: (v[:M] if k == "res" else v) for k, v in kw.items()}
    ops.gemm(A[:M], W, out[:M], code, **kws)                                   # mid kernel
    ops.gemm(A, W, big, code, **kw)                                            # tile kernels on 700 rows
    close_bf16(out[:M], ref, f"mid {epi}")
    assert torch.equal(out[:M], big[:M]), "a row's result must not depend on the number of rows in the call"
    ops.gemm(A[:M], W, out_ws[:M], code, workspace=torch.empty(64 << 20, dtype=torch.uint8, device=dev), **kws)
    close_bf16(out_ws[:M], ref, f"mid {epi} (K-sliced)")


@pytest.mark.parametrize("Tn,N,K,pad", [(4736, 4096, 4096, 0), (300, 1152, 2176, 0), (261 * 4, 1024, 4096, 0),
                                        (9472, 5632, 4096, 0), (1000, 512, 768, 64), (64, 256, 256, 0), (37, 264, 520, 8),
                                        (4176, 3072, 1024, 0), (4176, 1152, 4304, 0),
                                        # 288 tiles: persistent walk without a workspace (even K-tile count, ragged last
                                        # K-tile) / one tile per workgroup (odd K-tile count)
                                        (1000, 2304, 8192, 0), (1044, 4608, 4096, 0)])
def test_gemm_tn_matches_fp32_matmul(dev, Tn, N, K, pad):
    """bl_gemm_tn_bf16: out[N, K] = dyᵀ·x read untransposed (weight gradient of nn.Linear; torch autograd's
    `grad_output.t().mm(input)`). Ragged token counts, partial 256-tiles, column-slice views (ld > cols) and the K-split
    last round (5632 x 4096: 352 tiles) against the fp32 product of the same bf16 values."""
    from bridgelang_amd import train_ops as T
    g = torch.Generator(device="cpu").manual_seed(Tn + N)
    dyw = (torch.randn(Tn, N + pad, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    xw = torch.randn(Tn, K + pad, generator=g).to(torch.bfloat16).to(dev)
    dy, x = dyw[:, pad:], xw[:, :K]
    out = torch.full((N, K), float("nan"), dtype=torch.float32, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    T.gemm_tn(dy, x, out, workspace=ws)
    want = dy.float().t() @ x.float()
    err = (out - want).abs().max().item()
    scale = want.abs().max().item()
    print(f"T={Tn} N={N} K={K}: max err {err:.3e} of {scale:.3e}")
    assert torch.isfinite(out).all()
    assert err <= 2e-5 * scale + 1e-4
    out2 = torch.zeros_like(out)
    T.gemm_tn(dy, x, out2)                      # no workspace: same tiles in one launch
    assert (out2 - want).abs().max().item() <= 2e-5 * scale + 1e-4
