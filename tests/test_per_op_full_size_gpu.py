"""Per-kernel exactness at full 7B size on REAL activations (both synthetic checkpoints): every launch of the first
blocks of the DINOv2 tower, the SigLIP tower and the Llama decoder is re-computed by the CPU oracle (oracle/restate.py)
on the kernel's OWN input tensor, so each comparison isolates one kernel at its production shape (M = 261 / 256 / 288
rows, K up to 11008, head_dim 64 / 72 / 128) — LayerNorm / RMSNorm, GEMM + bias / GELU / LayerScale / residual / SwiGLU
epilogues, whole-sequence attention with and without fused RoPE + KV-cache write.

Bar (north_star: 1e-3 bf16 tolerance): per op, >= 99.9 % of the output elements are BIT-IDENTICAL to the oracle, the
relative rms error is <= 1e-4 and no element is off by more than one bf16 ulp of the tensor's largest element (2^-7
relative). Measured: 99.95–100 % bit-identical, rms 0–6e-5 (printed). The only legitimate difference is fp32 summation
order, which flips an output's bf16 rounding when the fp32 value sits within ~1e-6 of a rounding boundary.

End to end the same network turns these rare 1-ulp flips into percent-level logit differences (tests/
test_cfg_7b_golden_gpu.py, tools/drift_7b.py: the relative rms difference of the residual stream grows ~0.1 % per layer):
that is the sensitivity of a randomly initialised 32-layer decoder, not kernel error — which is why parity is stated per
kernel here and as ids / bounded logits end to end there.
"""
import pytest
import torch

from oracle import restate as R
from oracle import synth as S

pytestmark = pytest.mark.gpu
BLOCKS = 2


@pytest.mark.parametrize("recipe,model", [("decisive", "7b"), ("init", "7b"), ("init", "13b")])
def test_every_kernel_of_the_first_blocks_vs_oracle_full_size(dev, recipe, model):
    """model = "13b": BASELINE configs[4]'s composition (dinosiglip-vit-so-224px + llama2-13b-pure: hidden 5120, 40 heads,
    inter 13824) — the decoder kernels at 13B widths (K = 5120 / 13824 GEMMs, 40-head attention); the towers are the 7B ones."""
    from bridgelang_amd import ops, weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    from test_cfg_7b_golden_gpu import _weights
    from test_full_size_gpu import make_inputs
    dims, w = _weights(recipe, dev, model)
    eng = OpenVLAEngine(w, 1, 32)
    towers = ((w.dino, eng.dino_ops, eng.vbuf[0]), (w.siglip, eng.siglip_ops, eng.vbuf[1])) if model == "7b" else ()
    ids, pv = make_inputs(1, 32, 0)
    eng.set_inputs(ids.to(dev), pv.to(dev))
    specs = {s.name: s for s in W.tensor_specs(dims, recipe)}
    P = R.Prec(True)
    rows = []

    def T(name):
        s = specs[name]
        return S.synth_bf16(tuple(s.shape), S.tensor_seed(name, 0), s.mean, s.std).float()

    def show(tag, got, ref):
        got = got.float().cpu().reshape(ref.shape)
        d = (got - ref).abs()
        rows.append((tag, (got == ref).float().mean().item(), d.max().item() / ref.abs().max().item(),
                     (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))

    cpu = lambda t: t.float().cpu()
    for tw, plan, vb in towers:
        t = tw.dims
        Tn, Dm, Hp, hd = t.tokens, t.dim, t.mlp_pad, t.head_dim
        x = vb["x"][:Tn * Dm].view(1, Tn, Dm); h = vb["h"][:Tn * Dm].view(1, Tn, Dm); ao = vb["ao"][:Tn * Dm].view(1, Tn, Dm)
        qkv = vb["qkv"][:Tn * 3 * Dm].view(1, Tn, 3 * Dm); mlp = vb["mlp"][:Tn * Hp].view(1, Tn, Hp)
        n_pre = 3 if tw.prefix is not None else 2
        ops.run_all(plan[:n_pre]); torch.cuda.synchronize()
        pass
        for i in range(BLOCKS):
            b = lambda n: T(f"{t.prefix}.blocks.{i}.{n}")
            o7 = plan[n_pre + 7 * i:n_pre + 7 * (i + 1)]
            x_in = cpu(x)
            o7[0].run(); torch.cuda.synchronize()
            show(f"blk{i} norm1", h, R.layernorm(P, x_in, b("norm1.weight"), b("norm1.bias"), 1e-6)); h_in = cpu(h)
            o7[1].run(); torch.cuda.synchronize()
            show(f"blk{i} qkv gemm+bias", qkv, R.linear(P, h_in, b("attn.qkv.weight"), b("attn.qkv.bias"))); q_in = cpu(qkv)
            o7[2].run(); torch.cuda.synchronize()
            q3 = q_in.view(1, Tn, 3, t.heads, hd).permute(2, 0, 3, 1, 4)
            show(f"blk{i} attention hd={hd}", ao, R.attention(P, q3[0], q3[1], q3[2], hd ** -0.5, False).permute(0, 2, 1, 3).reshape(1, Tn, Dm)); a_in = cpu(ao)
            o7[3].run(); torch.cuda.synchronize()
            o = R.linear(P, a_in, b("attn.proj.weight"), b("attn.proj.bias"))
            if t.layerscale: o = P.rb(o * b("ls1.scale_factor"))
            show(f"blk{i} proj(+ls)+res", x, P.rb(x_in + o)); x1 = cpu(x)
            o7[4].run(); torch.cuda.synchronize()
            show(f"blk{i} norm2", h, R.layernorm(P, x1, b("norm2.weight"), b("norm2.bias"), 1e-6)); h_in = cpu(h)
            o7[5].run(); torch.cuda.synchronize()
            show(f"blk{i} fc1+gelu", mlp[..., :t.mlp], R.gelu(P, R.linear(P, h_in, b("mlp.fc1.weight"), b("mlp.fc1.bias")))); m_in = cpu(mlp[..., :t.mlp])
            o7[6].run(); torch.cuda.synchronize()
            o = R.linear(P, m_in, b("mlp.fc2.weight"), b("mlp.fc2.bias"))
            if t.layerscale: o = P.rb(o * b("ls2.scale_factor"))
            show(f"blk{i} fc2(+ls)+res", x, P.rb(x1 + o))
    # Llama: finish vision + projector, then the first layers op by op
    eng.run_vision(); ops.run_all(eng.projector_ops + eng.prefill_ops[:1]); torch.cuda.synchronize()
    D, H, hd, Sq = dims.llm_dim, dims.llm_heads, dims.head_dim, eng.S
    cos, sin = R.rope_tables(hd, dims.max_pos, dims.rope_theta)
    pass
    for l in range(BLOCKS):
        g = lambda n: T(f"language_model.model.layers.{l}.{n}")
        o7 = eng.prefill_ops[1 + 7 * l:1 + 7 * (l + 1)]
        x_in = cpu(eng.x)
        o7[0].run(); torch.cuda.synchronize()
        show(f"L{l} rmsnorm1", eng.h.view(1, Sq, D), R.rmsnorm(P, x_in, g("input_layernorm.weight"), dims.rms_eps)); h_in = cpu(eng.h.view(1, Sq, D))
        o7[1].run(); torch.cuda.synchronize()
        wq = torch.cat([g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight"), g("self_attn.v_proj.weight")])
        show(f"L{l} qkv gemm", eng.qkv.view(1, Sq, 3 * D), R.linear(P, h_in, wq)); q_in = cpu(eng.qkv.view(1, Sq, 3 * D))
        o7[2].run(); torch.cuda.synchronize()
        q3 = q_in.view(1, Sq, 3, H, hd).permute(2, 0, 3, 1, 4)
        qr, kr = R.apply_rope(P, q3[0], cos, sin, 0), R.apply_rope(P, q3[1], cos, sin, 0)
        show(f"L{l} rope(k) -> cache", eng.k_cache[l][:, :, :Sq], kr)
        show(f"L{l} attention hd=128 causal", eng.ao.view(1, Sq, D), R.attention(P, qr, kr, q3[2], hd ** -0.5, True).transpose(1, 2).reshape(1, Sq, D)); a_in = cpu(eng.ao.view(1, Sq, D))
        o7[3].run(); torch.cuda.synchronize()
        show(f"L{l} o_proj+res", eng.x, P.rb(x_in + R.linear(P, a_in, g("self_attn.o_proj.weight")))); x1 = cpu(eng.x)
        o7[4].run(); torch.cuda.synchronize()
        show(f"L{l} rmsnorm2", eng.h.view(1, Sq, D), R.rmsnorm(P, x1, g("post_attention_layernorm.weight"), dims.rms_eps)); h_in = cpu(eng.h.view(1, Sq, D))
        o7[5].run(); torch.cuda.synchronize()
        ga, up = R.linear(P, h_in, g("mlp.gate_proj.weight")), R.linear(P, h_in, g("mlp.up_proj.weight"))
        show(f"L{l} gate/up+swiglu", eng.act.view(1, Sq, -1), P.rb(P.rb(torch.nn.functional.silu(ga)) * up)); c_in = cpu(eng.act.view(1, Sq, -1))
        o7[6].run(); torch.cuda.synchronize()
        show(f"L{l} down+res", eng.x, P.rb(x1 + R.linear(P, c_in, g("mlp.down_proj.weight"))))

    print(f"\n{recipe} {model}: per-op parity at full width on real activations (op, bit-equal fraction, max|d|/max, rms(d)/rms)")
    for tag, eq, mx, rms in rows:
        print(f"  {tag:34s} {eq:.4f}  {mx:.2e}  {rms:.2e}")
    assert len(rows) == len(towers) * BLOCKS * 7 + BLOCKS * 8
    for tag, eq, mx, rms in rows:
        assert eq >= 0.999 and rms <= 1e-4 and mx <= 2 ** -7, (tag, eq, mx, rms)
