"""FP8 (e4m3) GEMM family (BASELINE configs[4]; no reference counterpart): the kernel is checked EXACTLY against a CPU
evaluation of the same quantised operands (products of e4m3 values are exact in fp32; only the summation order differs),
the quantiser against torch's e4m3 cast, and the quantisation error against the bf16 GEMM at the tolerance stated here."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def decode(codes: torch.Tensor) -> torch.Tensor:
    return codes.cpu().view(torch.float8_e4m3fn).float()


def test_quantize_rows_fp8(dev):
    from bridgelang_amd import ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, 1024, generator=g) * torch.logspace(-3, 2, 37)[:, None]).to(torch.bfloat16)
    x[5] = 0
    q, s, _ = ops.quantize_rows_fp8(x.to(dev))
    amax = x.float().abs().amax(dim=1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.equal(s.cpu(), want_s)
    inv = torch.where(amax > 0, torch.full_like(amax, 448.0) / amax, torch.ones_like(amax))   # (`448.0 / t` is t.reciprocal() * 448 in torch)
    want_q = (x.float() * inv[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    same = (q.cpu() == want_q).float().mean().item()
    assert same == 1.0, f"hardware RNE conversion vs torch cast: {same * 100:.3f}% identical codes"
    back = decode(q) * s.cpu()[:, None]
    assert ((back - x.float()).abs() <= x.float().abs() * 2 ** -4 + amax[:, None] * 2 ** -9 + 1e-30).all()   # 3 mantissa bits


@pytest.mark.parametrize("M,N,K,epi", [(4608, 4096, 4096, "res"), (300, 528, 256, "none"), (512, 1024, 1152 + 128, "swiglu"),
                                       (1000, 272, 1024, "f32"), (256, 256, 128, "bias")])
def test_gemm_fp8_exact(dev, M, N, K, epi):
    from bridgelang_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    A8, sa, _ = ops.quantize_rows_fp8(a.to(dev))
    W8, sw = ops.quantize_weight_fp8(w.to(dev))
    # the weight codes the kernel will see: undo the packing through the bf16 unpack
    wq = (w.float() / sw.cpu()[:, None]).to(torch.float8_e4m3fn).float()
    ref = (decode(A8).double() @ wq.double().T) * sa.cpu().double()[:, None] * sw.cpu().double()[None, :]
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16)
    rb = lambda t: t.float().to(torch.bfloat16).float()
    if epi == "f32":
        out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        ops.gemm_fp8(A8, sa, W8, sw, out, ops.EPI_F32)
        err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 1e-4, err          # fp32 accumulation inside / across the 128-wide MFMA steps (measured 1.9e-5 of the largest output)
        return
    if epi == "swiglu":   # rows 2j / 2j+1 of W are gate_j / up_j
        out = torch.zeros(M, N // 2, dtype=torch.bfloat16, device=dev)
        ops.gemm_fp8(A8, sa, W8, sw, out, ops.EPI_SWIGLU)
        gt, up = rb(ref[:, 0::2]), rb(ref[:, 1::2])
        want = rb(rb(torch.nn.functional.silu(gt)) * up)
    else:
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        kw = {"res": res.to(dev)} if epi == "res" else ({"bias": bias.to(dev)} if epi == "bias" else {})
        code = {"res": ops.EPI_RES, "bias": ops.EPI_BIAS, "none": ops.EPI_NONE}[epi]
        ops.gemm_fp8(A8, sa, W8, sw, out, code, **kw)
        want = rb(rb(ref.float()) + res.float()) if epi == "res" else (rb(ref.float() + bias.float()) if epi == "bias" else rb(ref.float()))
    got = out.cpu().float()
    mag = want.abs() + (ref.float().abs() if epi in ("res", "bias") else 0.0)      # an ulp of the rounded product, not only of the sum
    tol = mag * 2 ** -7 + want.abs().max() * 2 ** -12               # one bf16 ulp of slack for the fp32 summation order
    bad = ((got - want).abs() > tol).float().mean().item()
    # SwiGLU chains three roundings through the device's exp2/rcp silu (≈ 3 fp32 ulp): a flipped intermediate rounding may
    # move a handful of outputs by two bf16 ulps
    assert bad <= (1e-4 if epi == "swiglu" else 0.0), f"{bad * 100:.4f}% of outputs off by more than a bf16 ulp"
    assert ((got - want).abs() <= 4 * tol).all()
    exact = (got == want).float().mean().item()
    assert exact >= 0.97, f"only {exact * 100:.2f}% bit-identical to the CPU evaluation"


def test_gemm_fp8_vs_bf16_quantisation_error(dev):
    """End-to-end quantisation error of W8A8 (per-token × per-channel scales) against the bf16 GEMM on Gaussian data:
    relative Frobenius error ≈ 2^-4 / sqrt(3) · sqrt(2) ≈ 4 %; the bound stated for this path is 6 %."""
    from bridgelang_amd import ops
    M, N, K = 1024, 2048, 4096
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    ref = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a, ops.pack_weight(w), ref, ops.EPI_F32)
    A8, sa, _ = ops.quantize_rows_fp8(a)
    W8, sw = ops.quantize_weight_fp8(w)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ops.gemm_fp8(A8, sa, W8, sw, out, ops.EPI_F32)
    rel = ((out - ref).norm() / ref.norm()).item()
    print(f"\nW8A8 e4m3 vs bf16 GEMM: relative Frobenius error {rel:.4f}")
    assert rel <= 0.06


def test_gemm_fp8_rejects(dev):
    from bridgelang_amd import ops, _lib
    A8 = torch.zeros(64, 192, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        ops.quantize_weight_fp8(torch.zeros(64, 192, dtype=torch.bfloat16, device=dev))      # K % 128
    W8, sw = ops.quantize_weight_fp8(torch.ones(64, 256, dtype=torch.bfloat16, device=dev))
    with pytest.raises(ValueError):
        ops.gemm_fp8(A8, torch.ones(64, device=dev), W8, sw, torch.zeros(64, 64, dtype=torch.bfloat16, device=dev))


def test_engine_fp8_prefill_close_to_bf16(dev):
    """fp8=True engine (W8A8 prefill projections) vs the bf16 engine on a reduced-width model with Llama-sized channels
    (multiples of 128): first-token logits agree to quantisation noise; the stated bound is 8 % of the logit scale."""
    import dataclasses
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    from test_engine_gpu import make_inputs
    dims = W.tiny_dims()
    if dims.llm_dim % 128 or dims.llm_inter % 128 or dims.head_dim != 128:
        dims = dataclasses.replace(dims, llm_dim=512, llm_heads=4, head_dim=128, llm_inter=1536)
    w = W.allocate(dims, dev).fill_synthetic(seed=5)
    B, L = 3, 9
    ids, pv = make_inputs(dims, B, L, seed=1)
    e16, e8 = OpenVLAEngine(w, B, L), OpenVLAEngine(w, B, L, fp8=True)
    a = e16.generate(ids.to(dev), pv.to(dev)).clone()
    la = e16.logits[0].clone()
    b = e8.generate(ids.to(dev), pv.to(dev)).clone()
    lb = e8.logits[0].clone()
    assert any(op.name == "bl_gemm_fp8" for op in e8.prefill_ops) and not any(op.name == "bl_gemm_fp8" for op in e16.prefill_ops)
    rel = ((la - lb).abs().amax() / la.abs().amax()).item()
    print(f"\nfp8 prefill vs bf16: max |dlogit| / scale = {rel:.4f}; first tokens equal: {(a[:, 0] == b[:, 0]).float().mean().item():.2f}")
    assert 0 < rel <= 0.08


@pytest.mark.parametrize("T_,C_", [(592, 256), (9472 // 4, 1024), (130, 64), (257, 208)])
def test_transposing_quantise_equals_the_five_pass_form(dev, T_, C_):
    """The weight-gradient operands of the e4m3 path: bl_colamax_bf16 + bl_transpose_quantize_fp8 (read the bf16 [tokens,
    channels] matrix once, write token-contiguous e4m3 codes — row-major for the GEMM's activation side, fragment-major
    packed for its weight side) must reproduce round 3's transpose → quantise-rows → pack chain bit for bit: same
    per-channel scales (amax / 448), same RNE codes, zero padding of the token tail."""
    from bridgelang_amd import ops, train_ops as T
    from conftest import rand_bf16
    x = (rand_bf16((T_, C_), 3, 1.0) * torch.linspace(0.01, 30.0, C_)).to(torch.bfloat16).float()
    x[:, 5] = 0.0                                                  # an all-zero channel: scale 1, codes 0
    X = x.to(torch.bfloat16).to(dev)
    Tq = (T_ + 127) // 128 * 128
    # reference chain
    tX = torch.zeros(C_, Tq, dtype=torch.bfloat16, device=dev)
    T.transpose_pad(X, tX, Tq)
    q_ref, s_ref, _ = ops.quantize_rows_fp8(tX)
    # fused
    amax = torch.zeros(C_, dtype=torch.float32, device=dev)
    T.colamax(X, amax)
    assert torch.equal(amax.cpu(), x.abs().amax(0))
    q = torch.full((C_ * Tq,), 0x55, dtype=torch.uint8, device=dev)
    s = torch.zeros(C_, dtype=torch.float32, device=dev)
    T.transpose_quantize_fp8(X, amax, q, s, Tq, False)
    assert torch.equal(s, s_ref) and s[5].item() == 1.0
    assert torch.equal(q.view(C_, Tq), q_ref), f"{(q.view(C_, Tq) != q_ref).sum().item()} codes differ"
    if C_ % 16 == 0:
        p_ref = torch.zeros(C_ // 16, Tq // 64, 64, 8, dtype=torch.bfloat16, device=dev)
        T.pack(q_ref.view(torch.bfloat16), p_ref)
        qp = torch.full((C_ * Tq,), 0x55, dtype=torch.uint8, device=dev)
        T.transpose_quantize_fp8(X, amax, qp, s, Tq, True)
        assert torch.equal(qp.view(torch.bfloat16).view(p_ref.shape).view(torch.int16), p_ref.view(torch.int16))
