"""Full-size (openvla-7b, batch 16, S = 288) check of the training path through a size-independent property: the
directional derivative. With LoRA's B = 0 initialisation a step B ← −ε·g_B/‖g_B‖ is exactly representable, so
loss(B) − loss(0) must equal −ε‖g_B‖ to first order, and g_A must be exactly zero. This validates, at 7B, the whole
forward (K-concatenated adapters), the hand-written backward down to the vision towers, and the adapter gradients —
none of which the CPU oracle can reach at this size."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _free_gpu_memory():
    """These tests allocate tens to hundreds of GB: drop whatever earlier modules left in reference cycles / the caching
    allocator, and the synthetic checkpoints test_cfg_7b_golden_gpu keeps for its own parametrisations (15 GB each)."""
    import sys
    cached = getattr(sys.modules.get("test_cfg_7b_golden_gpu"), "_W", None)
    if cached:
        cached.clear()
    gc.collect()
    torch.cuda.empty_cache()
    yield
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("model,B", [("7b", 16), ("13b", 8)])
def test_lora_directional_derivative_full_size(dev, model, B):
    """`13b`: the BASELINE configs[4] composition (Llama-2-13B widths: hidden 5120, 40 heads, inter 13824, 40 layers) —
    the same property through the same training plan, which is what runs cfg 5's LoRA / per-rank step."""
    from bridgelang_amd import ops
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, openvla_7b_dims, prism_13b_dims
    L = 32
    w = allocate(openvla_7b_dims() if model == "7b" else prism_13b_dims(), dev).fill_synthetic(seed=0)
    lora = LoraAdapters(w, r=32)
    ts = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"), weight_decay=0.01)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, 31000, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
    ids[:, -1] = 2
    labels = torch.full((B, L), -100)
    labels[:, -8:] = ids[:, -8:]
    pv = torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)
    ts.set_batch(ids, None, pv, labels)
    loss0 = ts.forward().item()
    ts.backward()
    st = ts.store
    n = len(lora.adapters)
    gA = torch.stack([st.grad_view(f"lora.{i}.A").abs().max() for i in range(n)]).max().item()
    nB = torch.stack([st.grad_view(f"lora.{i}.B").double().pow(2).sum() for i in range(n)]).sum().sqrt().item()
    assert gA == 0.0, "dA must vanish while B = 0"
    assert nB > 0 and all(st.grad_view(f"lora.{i}.B").abs().max() > 0 for i in range(n)), "every adapter receives a gradient"
    for target in (0.1, 0.3):
        eps = target / nB
        for i, ad in enumerate(lora.adapters):
            ad.B.copy_((-(eps / nB) * st.grad_view(f"lora.{i}.B").view(ad.B.shape)).to(torch.bfloat16))
        ops.run_all(ts._adapter_ops)
        d = ts.forward().item() - loss0
        print(f"directional derivative: predicted {-target:.4f}, measured {d:.4f}")
        assert abs(d + target) <= 0.08 * target + 0.004, (target, d)
    del ts, lora, w
    torch.cuda.empty_cache()


def test_full_finetune_wgrad_consistent_with_lora_full_size(dev):
    """At B = 0 the adapter gradient of a linear is a projection of its full weight gradient: dB = s·dyᵀ·(x Aᵀ) =
    s·G_W·Aᵀ. The full fine-tuning step (stage vla-train: big-tile TN wgrad GEMMs `bl_gemm_tn_bf16` reading dy / x
    untransposed, fp32 accumulation into the flat gradient) and the LoRA step (small-output TN GEMMs
    `bl_gemm_tn_small_bf16`) compute the two sides through disjoint kernels; with the LoRA side pinned by the directional
    derivative above, agreement validates the 7B full-fine-tuning gradients of the sampled linears."""
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, openvla_7b_dims
    B, L = 16, 32
    dims = openvla_7b_dims()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 31000, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
    ids[:, -1] = 2
    labels = torch.full((B, L), -100)
    labels[:, -8:] = ids[:, -8:]
    pv = torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)
    names = ["language_model.model.layers.31.mlp.down_proj.weight", "language_model.model.layers.16.self_attn.o_proj.weight",
             "language_model.model.layers.0.self_attn.k_proj.weight", "language_model.model.layers.7.mlp.up_proj.weight",
             "projector.fc3.weight", "projector.fc2.weight"]
    w = allocate(dims, dev).fill_synthetic(seed=0)
    # optimizer state only for the sampled tensors (their fused groups): the backward chain is the full one
    from bridgelang_amd.training.step import ParamStore
    ts = TrainStep(w, "vla-train", B, L, store=ParamStore(w, "vla-train", only=names))
    ts.set_batch(ids, None, pv, labels)
    loss_full = ts.forward().item()
    ts.backward()
    GW = {n: ts.store.named_grad(n).float().clone() for n in names}
    del ts
    gc.collect()
    torch.cuda.empty_cache()
    lora = LoraAdapters(w, r=32, seed=3)
    tl = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"))
    tl.set_batch(ids, None, pv, labels)
    loss_lora = tl.forward().item()
    tl.backward()
    assert abs(loss_full - loss_lora) <= 2e-3 * loss_full          # B = 0: same function
    grads = lora.state_dict({u.key: tl.store.grad[u.offset:u.offset + u.numel] for u in tl.store.units})
    A = lora.state_dict()
    for n in names:
        mod = n[:-len(".weight")]
        a = A[f"base_model.model.{mod}.lora_A.weight"].to(dev)                       # [r, in]
        want = lora.scaling * (GW[n] @ a.t())                                          # [out, r]
        got = grads[f"base_model.model.{mod}.lora_B.weight"].to(dev)
        c = torch.nn.functional.cosine_similarity(got.flatten().double(), want.flatten().double(), dim=0).item()
        rel = ((got - want).norm() / want.norm()).item()
        print(f"{mod}: cosine {c:.5f}, rel. diff {rel:.4f}")
        assert c > 0.995 and rel < 0.1, (n, c, rel)
    del tl, lora, w
    torch.cuda.empty_cache()


def test_cfg3_vla_full_train_own_stage_full_size(dev):
    """BASELINE configs[2] in its OWN stage at full size: openvla-7b `vla-full-train` (vision towers trainable,
    prismatic/conf/vla.py:81-91), the per-GPU shape 32 × (256 patches + 40 text positions) = 32 × 296. No oracle reaches this
    size, so the step is checked through size-independent properties:
      (i)  the VISION-side weight gradients — hd 64 / 72 attention backward at 16 heads × 1024 / 1152, LayerScale / GELU /
           LayerNorm backward, the K-split TN wgrads at M = 8352 / 8192 — agree with the LoRA step's adapter gradients through
           dB = s·G_W·Aᵀ (two disjoint kernel paths; the LoRA side is pinned by the directional derivative above), on a DINOv2
           qkv, a SigLIP fc1, a DINOv2 fc2 and a SigLIP proj, beside two decoder linears at this batch;
      (ii) every loss and gradient norm is finite and the loss FALLS monotonically over five optimizer steps on one batch
           (forward, backward, clip, AdamW and the re-pack of both weight layouts all act on the same 7.5 B parameters)."""
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, openvla_7b_dims
    B, L = 32, 40
    dims = openvla_7b_dims()
    (ids, labels, pv), = _dummy_batches(B, L, 1, seed=11)
    names = ["vision_backbone.featurizer.blocks.2.attn.qkv.weight", "vision_backbone.fused_featurizer.blocks.5.mlp.fc1.weight",
             "vision_backbone.featurizer.blocks.20.mlp.fc2.weight", "vision_backbone.fused_featurizer.blocks.24.attn.proj.weight",
             "language_model.model.layers.3.self_attn.q_proj.weight", "language_model.model.layers.30.mlp.down_proj.weight"]
    w = allocate(dims, dev).fill_synthetic(seed=0)
    ts = TrainStep(w, "vla-full-train", B, L, max_grad_norm=1.0, weight_decay=0.0)
    assert ts.S == 296 and ts.train_vision and ts.store.n_params > 7.5e9
    ts.set_batch(ids, None, pv, labels)
    loss_full = ts.forward().item()
    ts.backward()
    GW = {n: ts.store.named_grad(n).float().clone() for n in names}
    assert all(torch.isfinite(g).all() and g.abs().max() > 0 for g in GW.values())
    # lr: 2e-5 (the reference's) overshoots on this random-init checkpoint with AdamW's unit-sized first steps
    # (11.04 -> 16.04 -> 11.60 -> 8.94 ...), 5e-6 descends monotonically (tools/explore_cfg3_lr.py); a first-order
    # "-lr·||g||_1" check is not available here: updates below half a bf16 ulp of a weight are absorbed by the bf16 copies.
    log = []
    for _ in range(5):
        loss, norm = ts.step(5e-6)
        log.append((loss.item(), norm.item()))
    final = ts.forward().item()
    torch.cuda.synchronize()
    gib = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"\n7B vla-full-train B = {B} x S = {ts.S}: (loss, grad norm) per step {[(round(l, 4), round(n, 2)) for l, n in log]}, "
          f"then {final:.4f}; {ts.store.n_params / 1e9:.2f} B trainable parameters, peak {gib:.0f} GiB")
    assert abs(log[0][0] - loss_full) <= 1e-6 * loss_full
    assert all(l == l and n == n and 0 < n < float("inf") for l, n in log) and final == final
    ls = [l for l, _ in log] + [final]
    assert all(b < a for a, b in zip(ls, ls[1:])), "the loss must fall on a repeated batch"
    assert final < log[0][0] - 1.5                                  # measured 11.04 -> 8.12 after five steps
    del ts
    gc.collect()
    torch.cuda.empty_cache()
    w.fill_synthetic(seed=0)                                        # the optimizer moved the weights: same checkpoint again
    lora = LoraAdapters(w, r=32, seed=3)
    tl = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"))
    tl.set_batch(ids, None, pv, labels)
    loss_lora = tl.forward().item()
    tl.backward()
    assert abs(loss_full - loss_lora) <= 2e-3 * loss_full          # B = 0: same function
    grads = lora.state_dict({u.key: tl.store.grad[u.offset:u.offset + u.numel] for u in tl.store.units})
    A = lora.state_dict()
    for n in names:
        mod = n[:-len(".weight")]
        a = A[f"base_model.model.{mod}.lora_A.weight"].to(dev)                       # [r, in]
        want = lora.scaling * (GW[n] @ a.t())                                          # [out, r]
        got = grads[f"base_model.model.{mod}.lora_B.weight"].to(dev)
        c = torch.nn.functional.cosine_similarity(got.flatten().double(), want.flatten().double(), dim=0).item()
        rel = ((got - want).norm() / want.norm()).item()
        print(f"{mod}: cosine {c:.5f}, rel. diff {rel:.4f}")
        assert c > 0.9995 and rel < 0.02, (n, c, rel)        # measured 0.99999 / 0.0052
    del tl, lora, w
    torch.cuda.empty_cache()


def test_cfg5_form_at_13b_widths(dev):
    """BASELINE configs[4]'s combination at the 13B layer widths (hidden 5120, 40 heads, inter 13824; four decoder layers so
    the optimizer state stays at 40 GB): full-shard parameter gathers + e4m3 forward / dgrad GEMMs + activation recomputation
    must reproduce the replicated-weight fp8 step bit for bit, step after step (the whole 40-layer model in this form is
    timed by `bench.py --mode train --model prism-13b --shard-params --fp8 --recompute`, profiles/)."""
    import dataclasses
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, prism_13b_dims
    dims = dataclasses.replace(prism_13b_dims(), llm_layers=4)
    B, L = 4, 32
    g = torch.Generator().manual_seed(1)
    batches = []
    for _ in range(2):
        ids = torch.randint(3, 31000, (B, L), generator=g)
        ids[:, 0] = 1
        ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
        ids[:, -1] = 2
        labels = torch.full((B, L), -100)
        labels[:, -8:] = ids[:, -8:]
        batches.append((ids, labels, torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)))
    out = {}
    for sharded in (False, True):
        w = allocate(dims, dev).fill_synthetic(seed=0)
        ts = TrainStep(w, "vla-train", B, L, max_grad_norm=1.0, weight_decay=0.1, fp8=True, shard_params=sharded,
                       recompute=sharded)
        log = []
        for ids, labels, pv in batches:
            ts.set_batch(ids, None, pv, labels)
            loss, norm = ts.step(1e-4)
            log.append((loss.item(), norm.item()))
        out[sharded] = (log, ts.store.full_master().cpu())
        assert w.layers_resident == (not sharded)
        del ts, w
        gc.collect()
        torch.cuda.empty_cache()
    print(out[False][0], out[True][0])
    assert all(l == l and n == n and n > 0 for l, n in out[True][0])                 # finite
    assert out[False][0] == out[True][0]
    assert torch.equal(out[False][1], out[True][1])


def _dummy_batches(B, L, n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        ids = torch.randint(3, 31000, (B, L), generator=g)
        ids[:, 0] = 1
        ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g)
        ids[:, -1] = 2
        labels = torch.full((B, L), -100)
        labels[:, -8:] = ids[:, -8:]
        out.append((ids, labels, torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)))
    return out


def test_cfg5_full_depth_13b_full_shard_fp8_recompute(dev):
    """BASELINE configs[4] in its OWN form at FULL depth: the 40-layer Llama-2-13B VLA (13.8 B parameters), full fine-tune,
    FSDP full-shard parameter gathers + e4m3 GEMMs + activation recomputation, one GPU's share of the step (the gather is
    a copy at world size 1). No oracle reaches this size; the step is checked through properties: every loss and
    gradient norm finite, the loss FALLS over three optimizer steps on one batch (forward, backward, clip, AdamW and the
    per-layer gather / re-quantise cycle all act on the same parameters), the model's decoder layers are sharded out during
    training and come back bit-consistent with the fp32 masters afterwards."""
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, prism_13b_dims
    dims = prism_13b_dims()
    assert dims.llm_layers == 40 and dims.llm_dim == 5120
    B, L = 4, 40
    (ids, labels, pv), = _dummy_batches(B, L, 1, seed=7)
    w = allocate(dims, dev).fill_synthetic(seed=0)
    ts = TrainStep(w, "vla-full-train", B, L, max_grad_norm=1.0, weight_decay=0.0, fp8=True, shard_params=True, recompute=True)
    assert not w.layers_resident and ts.store.n_params > 13.5e9
    ts.set_batch(ids, None, pv, labels)
    log = []
    for _ in range(3):
        loss, norm = ts.step(2e-5)
        log.append((loss.item(), norm.item()))
    torch.cuda.synchronize()
    gib = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"\n13B full-shard + fp8 + recompute, 40 layers, B = {B}: (loss, grad norm) per step {[(round(l, 4), round(n, 2)) for l, n in log]}, "
          f"{ts.store.n_params / 1e9:.2f} B trainable parameters, peak {gib:.0f} GiB")
    assert all(l == l and n == n and 0 < n < float("inf") for l, n in log)
    assert log[2][0] < log[1][0] < log[0][0], "the loss must fall on a repeated batch"
    ts.materialize_params()
    assert w.layers_resident
    lw = w.layers[39]
    from bridgelang_amd import ops
    got = ops.unpack_weight(lw.o_w).float()
    u = ts._layer_units[(39, "o_w")]                     # world 1: the rank's master slice is the whole flat space
    want = ts.store.master[u.offset:u.offset + u.numel].view(u.group.n, u.group.k).to(torch.bfloat16).float()
    assert torch.equal(got, want), "materialised layer weights must be the bf16 rounding of the fp32 masters"
    del ts, w
    gc.collect()
    torch.cuda.empty_cache()


def test_cfg5_fp8_gradient_norm_close_to_bf16_on_a_13b_prefix(dev):
    """The stated fp8 bound at 13B widths: on a 4-layer prefix of the 13B decoder the first step's loss agrees with the
    bf16 step within 2 % and the global gradient norm within 5 % (same weights, same batch; e4m3 forward / dgrad GEMMs vs
    bf16 — quantisation noise, not a bit-exact claim)."""
    import dataclasses
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, prism_13b_dims
    dims = dataclasses.replace(prism_13b_dims(), llm_layers=4)
    B, L = 4, 40
    (ids, labels, pv), = _dummy_batches(B, L, 1, seed=9)
    res = {}
    for fp8 in (False, True):
        w = allocate(dims, dev).fill_synthetic(seed=0)
        ts = TrainStep(w, "vla-train", B, L, max_grad_norm=1.0, fp8=fp8)
        ts.set_batch(ids, None, pv, labels)
        loss, norm = ts.step(1e-5)
        res[fp8] = (loss.item(), norm.item())
        del ts, w
        gc.collect()
        torch.cuda.empty_cache()
    (l0, n0), (l1, n1) = res[False], res[True]
    print(f"\n13B-width 4-layer prefix: bf16 loss {l0:.4f} norm {n0:.3f} | fp8 loss {l1:.4f} norm {n1:.3f}")
    assert abs(l1 - l0) <= 0.02 * abs(l0) and abs(n1 - n0) <= 0.05 * n0
