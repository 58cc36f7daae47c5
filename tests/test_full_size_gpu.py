"""Full-size (openvla-7b widths, batch 16, S = 288) GPU checks.

The CPU oracle cannot run the whole 7B model in test time, so full size is covered two ways:
  * oracle SPOT CHECKS at full width on real intermediate tensors of the engine: one Llama decoder layer (prefill,
    S = 288, K = 4096 / 11008 GEMMs incl. the split-K tail, causal attention at head_dim 128), the projector, and one
    block of each ViT tower — weights are copied back from the device arena (the packing is proven bit-exact at small
    size), inputs are the engine's own activations for 2 of the 16 sequences;
  * size-independent PROPERTIES of the whole 7B path: replay determinism, batch-permutation equivariance (a sequence's
    ids do not depend on its slot or its batch mates), duplicate sequences give duplicate ids, cached decode ≡ the ids
    fed back through a longer prefill (KV-cache consistency), the pipelined server ≡ the plain engine.
"""
import pytest
import torch

from oracle import restate as R

pytestmark = pytest.mark.gpu

B, L = 16, 32


def make_inputs(batch, seq, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 224, 224, 3), generator=g, dtype=torch.uint8).float().div_(255.0).permute(0, 3, 1, 2)
    m = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    s = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    pv = torch.cat([(img - m) / s, (img - 0.5) / 0.5], dim=1).to(torch.bfloat16)
    ids = torch.randint(3, 31743, (batch, seq), generator=g)
    ids[:, 0], ids[:, -1] = 1, 29871
    return ids, pv


@pytest.fixture(scope="module")
def full(dev):
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    dims = W.openvla_7b_dims()
    w = W.allocate(dims, dev).fill_synthetic(seed=0)
    eng = OpenVLAEngine(w, B, L)
    ids, pv = make_inputs(B, L, 0)
    return dims, w, eng, ids.to(dev), pv.to(dev)


def rel(got, ref):
    return ((got.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def test_llama_layer_full_width_vs_oracle(full, dev):
    """Layer 0 prefill at full width on the engine's own inputs (2 of 16 sequences checked on the CPU)."""
    from bridgelang_amd import ops
    from bridgelang_amd.weights import _unpack
    dims, w, eng, ids, pv = full
    eng.set_inputs(ids, pv)
    eng.run_vision()
    ops.run_all(eng.projector_ops + eng.prefill_ops[:1])          # … + embedding splice
    torch.cuda.synchronize()
    x_in = eng.x.clone()
    ops.run_all(eng.prefill_ops[1:9])                              # the 8 launches of decoder layer 0
    torch.cuda.synchronize()
    lw = w.layers[0]
    D, I = dims.llm_dim, dims.llm_inter
    qkv, gu = _unpack(lw.qkv_w).float().cpu(), _unpack(lw.gu_w).float().cpu()
    lm = "language_model.model.layers.0"
    sd = {f"{lm}.input_layernorm.weight": lw.ln1.float().cpu(), f"{lm}.post_attention_layernorm.weight": lw.ln2.float().cpu(),
          f"{lm}.self_attn.q_proj.weight": qkv[:D], f"{lm}.self_attn.k_proj.weight": qkv[D:2 * D],
          f"{lm}.self_attn.v_proj.weight": qkv[2 * D:], f"{lm}.self_attn.o_proj.weight": _unpack(lw.o_w).float().cpu(),
          f"{lm}.mlp.gate_proj.weight": gu[0::2], f"{lm}.mlp.up_proj.weight": gu[1::2],
          f"{lm}.mlp.down_proj.weight": _unpack(lw.down_w).float().cpu(),
          "language_model.model.norm.weight": torch.ones(D), "language_model.lm_head.weight": torch.zeros(16, D)}
    sel = [0, B - 1]
    p = R.Prec(True)
    xs = x_in[sel].float().cpu()
    # run the oracle layer by calling llama_forward with one layer and reading the residual stream back out of the cache
    # path: re-implement the residual output from its pieces
    h = R.rmsnorm(p, xs, sd[f"{lm}.input_layernorm.weight"], dims.rms_eps)
    H, hd, S = dims.llm_heads, dims.head_dim, eng.S
    q = R.linear(p, h, sd[f"{lm}.self_attn.q_proj.weight"]).view(2, S, H, hd).transpose(1, 2)
    k = R.linear(p, h, sd[f"{lm}.self_attn.k_proj.weight"]).view(2, S, H, hd).transpose(1, 2)
    v = R.linear(p, h, sd[f"{lm}.self_attn.v_proj.weight"]).view(2, S, H, hd).transpose(1, 2)
    cos, sin = R.rope_tables(hd, dims.max_pos, dims.rope_theta)
    q, k = R.apply_rope(p, q, cos, sin, 0), R.apply_rope(p, k, cos, sin, 0)
    ek = rel(eng.k_cache[0][sel][:, :, :S], k)
    assert ek < 8e-3, ek            # one bf16 ulp of the largest element is 3.9e-3
    a = R.attention(p, q, k, v, hd ** -0.5, True).transpose(1, 2).reshape(2, S, D)
    x1 = p.rb(xs + R.linear(p, a, sd[f"{lm}.self_attn.o_proj.weight"]))
    h2 = R.rmsnorm(p, x1, sd[f"{lm}.post_attention_layernorm.weight"], dims.rms_eps)
    act = p.rb(p.rb(torch.nn.functional.silu(R.linear(p, h2, sd[f"{lm}.mlp.gate_proj.weight"]))) *
               R.linear(p, h2, sd[f"{lm}.mlp.up_proj.weight"]))
    x2 = p.rb(x1 + R.linear(p, act, sd[f"{lm}.mlp.down_proj.weight"]))
    e = rel(eng.x[sel], x2)
    print(f"\nfull-width Llama layer 0: rel err of the residual stream {e:.3g}, of the rotated keys {ek:.3g}")
    assert e < 1e-2                 # measured 6.6e-3 (7 kernels chained); per-kernel bars: test_per_op_full_size_gpu.py


def test_projector_and_vit_block_full_width_vs_oracle(full, dev):
    from bridgelang_amd import ops
    from bridgelang_amd.weights import _unpack
    dims, w, eng, ids, pv = full
    eng.set_inputs(ids, pv)
    eng.run_vision()
    torch.cuda.synchronize()
    p = R.Prec(True)
    # projector on the engine's own fused features (2 sequences)
    feats = eng.feats.view(B, 256, -1)[[0, B - 1]].float().cpu()
    sd = {"projector.fc1.weight": _unpack(w.fc1_w).float().cpu(), "projector.fc1.bias": w.fc1_b.float().cpu(),
          "projector.fc2.weight": _unpack(w.fc2_w).float().cpu(), "projector.fc2.bias": w.fc2_b.float().cpu(),
          "projector.fc3.weight": _unpack(w.fc3_w).float().cpu(), "projector.fc3.bias": w.fc3_b.float().cpu()}
    ops.run_all(eng.projector_ops)
    torch.cuda.synchronize()
    e = rel(eng.x[[0, B - 1], 1:257], R.projector(p, sd, feats))
    print(f"\nfull-width projector rel err {e:.3g}")
    assert e < 8e-3                 # measured 5e-3 (3 GEMMs + 2 GELUs chained)
    # first block of each tower, re-run in isolation from a known input
    for ti, (tw, vb) in enumerate(((w.dino, eng.vbuf[0]), (w.siglip, eng.vbuf[1]))):
        t = tw.dims
        T, Dm = t.tokens, t.dim
        M = B * T
        plan = eng.dino_ops if ti == 0 else eng.siglip_ops
        n_pre = 3 if tw.prefix is not None else 2               # im2col (+ prefix) + patch-embed GEMM
        ops.run_all(plan[:n_pre])
        torch.cuda.synchronize()
        x0 = vb["x"][:M * Dm].view(B, T, Dm)[[0, B - 1]].float().cpu()
        ops.run_all(plan[n_pre:n_pre + 7])                       # block 0
        torch.cuda.synchronize()
        b = tw.blocks[0]
        g = lambda z: z.float().cpu()
        h = R.layernorm(p, x0, g(b.norm1_w), g(b.norm1_b), dims.ln_eps)
        qkv = R.linear(p, h, _unpack(b.qkv_w).float().cpu(), g(b.qkv_b)).view(2, T, 3, t.heads, t.head_dim).permute(2, 0, 3, 1, 4)
        a = R.attention(p, qkv[0], qkv[1], qkv[2], t.head_dim ** -0.5, False).permute(0, 2, 1, 3).reshape(2, T, Dm)
        o = R.linear(p, a, _unpack(b.proj_w).float().cpu(), g(b.proj_b))
        if b.ls1 is not None:
            o = p.rb(o * g(b.ls1))
        x1 = p.rb(x0 + o)
        h = R.layernorm(p, x1, g(b.norm2_w), g(b.norm2_b), dims.ln_eps)
        f = R.gelu(p, R.linear(p, h, _unpack(b.fc1_w).float().cpu()[:t.mlp], g(b.fc1_b)[:t.mlp]))
        o = R.linear(p, f, _unpack(b.fc2_w).float().cpu()[:, :t.mlp], g(b.fc2_b))
        if b.ls2 is not None:
            o = p.rb(o * g(b.ls2))
        x2 = p.rb(x1 + o)
        e = rel(vb["x"][:M * Dm].view(B, T, Dm)[[0, B - 1]], x2)
        print(f"full-width {t.prefix.split('.')[-1]} block 0 rel err {e:.3g}")
        assert e < (5e-3 if t.layerscale else 1.6e-2)      # measured 2.5e-3 (DINOv2, LayerScale 0.1) / 1.1e-2 (SigLIP)


def test_replay_determinism_and_batch_equivariance(full, dev):
    dims, w, eng, ids, pv = full
    a = eng.generate(ids, pv).clone()
    la = eng.logits.clone()
    b = eng.generate(ids, pv).clone()
    assert torch.equal(a, b) and torch.equal(la, eng.logits), "two runs on the same inputs must be bit-identical"
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    c = eng.generate(ids[perm], pv[perm]).clone()
    assert torch.equal(c, a[perm]), "a sequence's ids must not depend on its batch slot / batch mates"
    ids2, pv2 = ids.clone(), pv.clone()
    ids2[1], pv2[1] = ids2[0], pv2[0]                               # duplicate sequence 0 into slot 1
    d = eng.generate(ids2, pv2)
    assert torch.equal(d[0], d[1]) and torch.equal(d[0], a[0])
    assert a.min().item() >= 0 and a.max().item() < dims.vocab


def test_graph_and_pipeline_equal_eager_full_size(full, dev):
    from bridgelang_amd.pipeline import TwoStagePipeline
    dims, w, eng, ids, pv = full
    eager = eng.generate(ids, pv).clone()
    eng.capture()
    assert torch.equal(eng.generate(ids, pv), eager)
    eng._graph = None
    ids_b, pv_b = make_inputs(B, L, 1)
    want_b = eng.generate(ids_b.to(dev), pv_b.to(dev)).clone()
    pipe = TwoStagePipeline(w, B, L)
    for e in pipe.engines:
        e.set_inputs(ids, pv)
    pipe.capture()
    pipe.step(ids, pv)
    out_a = pipe.step(ids_b.to(dev), pv_b.to(dev)).clone()
    out_b = pipe.flush().clone()
    assert torch.equal(out_a, eager) and torch.equal(out_b, want_b)


@pytest.mark.parametrize("split", [False, True])
def test_staggered_pipeline_full_size(full, dev, split):
    """7 (8 with the vision stage split off) batches in flight at 7B width, 96 merged decode rows through
    bl_gemm_skinny_rows_bf16 + bl_rmsnorm_skinny_bf16: every sequence's logits at every decode iteration and all ids are
    bit-identical to the plain (oracle-checked) engine's — the path bench.py measures IS the verified path."""
    from bridgelang_amd.pipeline import StaggeredDecodePipeline
    dims, w, eng, ids, pv = full
    eng._graph = None
    n = 10
    batches = [(ids, pv)] + [tuple(t.to(dev) for t in make_inputs(B, L, 10 + s)) for s in range(n - 1)]
    want, want_lg = [], []
    for i, p in batches:
        want.append(eng.generate(i, p).clone().cpu())
        want_lg.append(eng.logits.clone())
    pipe = StaggeredDecodePipeline(w, B, L, split_vision=split)
    for e in pipe.engines:
        e.set_inputs(ids, pv)
    pipe.capture()
    got, checked = [], 0
    for k, (i, p) in enumerate(batches):
        out = pipe.step(i, p).clone()
        if k >= pipe.slots - 1:
            got.append(out.cpu())
        for g in range(1, pipe.n_new):       # this step ran decode iteration g of batch j
            j = k - g - pipe.lag
            if j < 0:
                continue
            a, b = pipe.logits[(g - 1) * B:g * B], want_lg[j][g]
            assert torch.equal(a, b), (f"step {k}: logits of batch {j}, decode iteration {g} differ from the engine's in "
                                       f"{(a != b).any(dim=1).sum().item()} of {B} sequences")
            checked += B
    got += [o.cpu() for o in pipe.flush()]
    assert len(got) == n and checked >= n * B * 2
    for k in range(n):
        assert torch.equal(got[k], want[k]), f"batch {k}: ids differ from the plain engine's"
    print(f"\nstaggered pipeline (split_vision={split}): {checked} (sequence, iteration) logit rows and {n * B} id rows "
          f"bit-identical to the plain engine")
    del pipe


def test_fp8_prefill_full_size(full, dev):
    """W8A8 e4m3 prefill projections at 7B width (31 of 32 layers, 124 fp8 GEMMs per prefill) against the bf16 engine:
    each GEMM carries ≈ 3.7 % quantisation noise (tests/test_fp8_gpu.py) and a random-init network neither damps it nor
    has decisive logits, so after 124 GEMMs the first-token logits keep a cosine of ≈ 0.96 with the bf16 ones (max
    deviation 20–31 % of the logit scale) and only ≈ 40 % of the synthetic sequences keep their greedy token. Stated
    bounds: cosine ≥ 0.95, deviation ≤ 40 % of the scale. No trained weights exist offline to measure the task-level effect."""
    from bridgelang_amd.engine import OpenVLAEngine
    dims, w, eng, ids, pv = full
    eng._graph = None
    a = eng.generate(ids, pv).clone()
    la = eng.logits[0].clone()
    e8 = OpenVLAEngine(w, B, L, fp8=True)
    b = e8.generate(ids, pv).clone()
    lb = e8.logits[0].clone()
    rel = ((la - lb).abs().amax(dim=1) / la.abs().amax(dim=1))
    cos = torch.nn.functional.cosine_similarity(la, lb, dim=1)
    print(f"\nfp8 prefill at 7B: max |dlogit| / scale per sequence {rel.min().item():.3f} … {rel.max().item():.3f}; "
          f"logit cosine {cos.min().item():.4f} … {cos.max().item():.4f}; first tokens equal {(a[:, 0] == b[:, 0]).float().mean().item():.2f}")
    assert rel.max().item() <= 0.4 and cos.min().item() >= 0.95
    del e8


@pytest.mark.parametrize("recipe", ["decisive", "init"])
def test_kv_cache_consistency_full_size(full, dev, recipe):
    """Greedy token t+1 produced by the cached decode must equal the greedy token produced by a fresh prefill over
    prompt + tokens[0..t] (the reference's use_cache=False path, run_openvla_demo.py:43, gives the same ids). The two
    paths are identical math in different kernels (weight-streaming GEMM / decode attention vs tiled GEMM / prefill
    attention: other fp32 summation orders), so logits agree to the network's bf16 noise and ids may differ only where
    the top-2 gap is inside that noise — on the "decisive" checkpoint that leaves most sequences in agreement; on the
    bench checkpoint ("init": flat logits, chaotic network) the agreement rate is printed, not asserted."""
    from bridgelang_amd.engine import OpenVLAEngine
    from test_cfg_7b_golden_gpu import _weights
    dims, w0, eng0, ids, pv = full
    if recipe == "init":
        w, eng = w0, eng0
    else:
        _, w = _weights(recipe, dev)
        eng = OpenVLAEngine(w, B, L)
    got = eng.generate(ids, pv).clone()
    for t in (1, 3):
        longer = torch.cat([ids, got[:, :t]], dim=1)
        e2 = OpenVLAEngine(w, B, L + t, n_new=1)
        nxt = e2.generate(longer, pv)[:, 0]
        lg_cached, lg_fresh = eng.logits[t], e2.logits[0]
        scale = lg_fresh.abs().max().item()
        dl = (lg_cached - lg_fresh).abs().amax(dim=1)                  # per-sequence logit noise between the two paths
        same = (nxt == got[:, t]).float().mean().item()
        print(f"\n{recipe}: cached step {t}: {same * 100:.0f}% of sequences agree with the uncached recomputation; "
              f"max |dlogit| {dl.max().item():.3g} (scale {scale:.3g})")
        assert dl.max().item() <= 4e-2 * scale                          # measured 2.8–3.1 % over input realisations
        for bidx in (nxt != got[:, t]).nonzero().flatten().tolist():
            top2 = lg_fresh[bidx].topk(2).values
            assert (top2[0] - top2[1]).item() <= 2 * dl[bidx].item() + 1e-6, f"sequence {bidx}: decisive gap"
        if recipe == "decisive":
            assert same >= 0.75
        del e2
