"""LoRA fine-tuning step (vla-scripts/finetune.py:174-189,253-318) vs autograd over the CPU oracle: adapters on every
nn.Linear except lm_head (vision qkv/proj/fc1/fc2, projector, q/k/v/o/gate/up/down), scaling alpha/r = 16/32, base frozen,
AdamW(lr, weight_decay=0.01 — torch default) on the adapters only."""
import pytest
import torch

from oracle import restate as R
from test_train_step_gpu import cos, make_batch, oracle_loss

pytestmark = pytest.mark.gpu


def random_adapters(lora, seed, b_std=0.02):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for ad in lora.adapters:
        for j, mod in enumerate(ad.modules):
            out, inp = ad.shapes[j]
            sd[f"base_model.model.{mod}.lora_A.weight"] = (torch.randn(lora.r, inp, generator=g) / lora.r).to(torch.bfloat16).float()
            sd[f"base_model.model.{mod}.lora_B.weight"] = (torch.randn(out, lora.r, generator=g) * b_std).to(torch.bfloat16).float()
    return sd


def effective_sd(sd, ad_sd, scaling, mods):
    """HF-named weights with W + s·B·A substituted (differentiable w.r.t. the adapter tensors)."""
    out = dict(sd)
    for mod in mods:
        a, b = ad_sd[f"base_model.model.{mod}.lora_A.weight"], ad_sd[f"base_model.model.{mod}.lora_B.weight"]
        out[mod + ".weight"] = sd[mod + ".weight"] + scaling * (b @ a)
    return out


def test_lora_structure_and_names(dev):
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.weights import allocate, tiny_dims, openvla_7b_dims
    dims = tiny_dims()
    w = allocate(dims, dev).fill_synthetic(seed=3)
    lora = LoraAdapters(w, r=32)
    assert lora.scaling == 0.5
    mods = [m for ad in lora.adapters for m in ad.modules]
    assert "language_model.lm_head" not in mods and not any("patch_embed" in m for m in mods)
    per_layer = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
    assert all(f"language_model.model.layers.0.{n}" in mods for n in per_layer)
    assert all(f"projector.fc{i}" in mods for i in (1, 2, 3))
    assert "vision_backbone.featurizer.blocks.0.attn.qkv" in mods and "vision_backbone.fused_featurizer.blocks.1.mlp.fc2" in mods
    sd = lora.state_dict()
    assert sd["base_model.model.language_model.model.layers.1.mlp.up_proj.lora_A.weight"].shape == (32, dims.llm_dim)
    assert sd["base_model.model.language_model.model.layers.1.mlp.up_proj.lora_B.weight"].shape == (dims.llm_inter, 32)
    assert all(v.abs().sum() == 0 for k, v in sd.items() if "lora_B" in k)            # gaussian init: B = 0
    a = sd["base_model.model.projector.fc2.lora_A.weight"]
    assert abs(a.std().item() - 1 / 32) < 0.1 / 32
    rt = LoraAdapters(w, r=32, seed=1).load_state_dict(sd).state_dict()               # round trip through the fused layout
    assert all(torch.equal(rt[k], sd[k]) for k in sd)
    # 7B parameter count of the adapters on the executed path (LLM 32 layers + projector + 23 + 26 vision blocks)
    d7 = openvla_7b_dims()
    r = 32
    llm = 32 * (4 * r * 2 * 4096 + 3 * r * (4096 + 11008))
    proj = r * ((2176 + 8704) + (8704 + 4096) + (4096 + 4096))
    dino = d7.dino.n_run * r * ((1024 + 3072) + 2048 + 2 * (1024 + 4096))
    sig = d7.siglip.n_run * r * ((1152 + 3456) + 2304 + 2 * (1152 + 4304))
    assert llm == 79_953_920 and llm + proj + dino + sig == 107_862_016


def test_lora_gradients_and_adamw_step(dev):
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, tiny_dims
    dims = tiny_dims()
    w = allocate(dims, dev).fill_synthetic(seed=3)
    sd = {k: v.float().cpu() for k, v in w.state_dict().items()}
    lora = LoraAdapters(w, r=32)
    ad_sd = random_adapters(lora, seed=7)
    lora.load_state_dict(ad_sd)
    mods = [m for ad in lora.adapters for m in ad.modules]
    B, L, lr = 3, 20, 5e-4
    ts = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"), weight_decay=0.01)
    assert ts.store.n_params == sum(ad.A.numel() + ad.B.numel() for ad in lora.adapters)
    ids, mask, labels, pv = make_batch(dims, B, L)
    ts.set_batch(ids, mask, pv, labels)
    loss = ts.forward().item()
    ts.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in ad_sd.items()}
    ref = oracle_loss(effective_sd(sd, params, lora.scaling, mods), dims, ids, mask, labels, pv)
    ref.backward()
    print(f"[lora] loss {loss:.5f} vs oracle {ref.item():.5f}; {len(params)} adapter tensors, {lora.n_params()} params")
    assert abs(loss - ref.item()) <= 3e-3 * abs(ref.item())
    masters = {u.key: ts.store.grad[u.offset:u.offset + u.numel] for u in ts.store.units}
    got = lora.state_dict(masters)                      # gradients, un-fused into PEFT names / shapes
    worst = 1.0
    for k, want in params.items():
        g, scale = got[k], want.grad.abs().max().item()
        c = cos(g, want.grad)
        worst = min(worst, c)
        assert scale > 0 and c > 0.985 and (g - want.grad).abs().max().item() <= 0.08 * scale, (k, c)
    print(f"[lora] worst adapter-gradient cosine {worst:.5f}")
    # off-block entries of the fused adapters carry no gradient
    for i, ad in enumerate(lora.adapters):
        if len(ad.modules) > 1:
            gB = masters[f"lora.{i}.B"].view(ad.B.shape).cpu()
            keep = torch.zeros_like(gB, dtype=torch.bool)
            for j in range(len(ad.modules)):
                keep[ad.member_rows(j), j * 64:(j + 1) * 64] = True
            assert (gB[~keep] == 0).all() and gB[keep].abs().sum() > 0
    # one AdamW step (no clipping in finetune.py; torch default weight decay on every adapter tensor)
    opt = torch.optim.AdamW(list(params.values()), lr=lr)
    opt.step()
    ts.clip_grad_norm()
    ts.optimizer_step(lr)
    full = ts.store.full_master()
    new = lora.state_dict({u.key: full[u.offset:u.offset + u.numel] for u in ts.store.units})
    live = lora.state_dict()
    for k, want in params.items():
        upd, upd_ref = new[k] - ad_sd[k], want.detach() - ad_sd[k]
        assert cos(upd, upd_ref) > 0.97, (k, cos(upd, upd_ref))
        assert torch.equal(live[k], new[k].to(torch.bfloat16).float()), k
    # second forward runs on the re-packed adapters: loss must move the way the oracle's does
    loss2 = ts.forward().item()
    ref2 = oracle_loss(effective_sd(sd, {k: v.detach().to(torch.bfloat16).float() for k, v in params.items()}, lora.scaling, mods),
                       dims, ids, mask, labels, pv)
    print(f"[lora] after one step: loss {loss2:.5f} vs oracle {ref2.item():.5f}")
    assert abs(loss2 - ref2.item()) <= 3e-3 * abs(ref2.item()) and loss2 < loss


def test_lora_three_step_loss_curve_vs_oracle(dev):
    """Three optimisation steps from the PEFT initialisation (A gaussian, B = 0): the loss curve must follow the CPU
    restatement trained with torch.optim.AdamW on bf16-rounded adapter copies (SURVEY §8d cfg 3/4: loss-curve parity)."""
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, tiny_dims
    dims = tiny_dims()
    w = allocate(dims, dev).fill_synthetic(seed=3)
    sd = {k: v.float().cpu() for k, v in w.state_dict().items()}
    lora = LoraAdapters(w, r=32, seed=5)
    init = lora.state_dict()
    assert all(v.abs().sum() == 0 for k, v in init.items() if "lora_B" in k)
    mods = [m for ad in lora.adapters for m in ad.modules]
    B, L, lr = 3, 20, 2e-3
    ts = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"), weight_decay=0.01)
    params = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=lr)
    got, want = [], []
    for step in range(3):
        ids, mask, labels, pv = make_batch(dims, B, L, seed=40 + step)
        ts.set_batch(ids, mask, pv, labels)
        loss, _ = ts.step(lr)
        got.append(loss.item())
        live = {k: v.detach().to(torch.bfloat16).float().requires_grad_(True) for k, v in params.items()}
        ref = oracle_loss(effective_sd(sd, live, lora.scaling, mods), dims, ids, mask, labels, pv)
        ref.backward()
        for k in params:
            params[k].grad = live[k].grad
        opt.step()
        opt.zero_grad()
        want.append(ref.item())
    print("lora loss curve", [round(x, 4) for x in got], "oracle", [round(x, 4) for x in want])
    assert all(abs(a - b) <= 5e-3 * abs(b) for a, b in zip(got, want))
    assert got[2] < got[0] - 0.02


def test_gradient_accumulation_equals_one_big_batch(dev):
    """Two micro-batches of 2 accumulated with loss / 2 (finetune.py:256-262) give the gradients of one batch of 4 (every
    sample carries 8 supervised tokens, so the mean of the micro-batch means is the big-batch mean)."""
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, tiny_dims
    dims = tiny_dims()
    w = allocate(dims, dev).fill_synthetic(seed=3)
    ids, mask, labels, pv = make_batch(dims, 4, 20, seed=9, ragged=False)
    grads = {}
    for mode in ("big", "accum"):
        lora = LoraAdapters(w, r=32, seed=2)
        lora.load_state_dict(random_adapters(lora, seed=4))
        Bm = 4 if mode == "big" else 2
        ts = TrainStep(w, "lora", Bm, 20, lora=lora, max_grad_norm=float("inf"), weight_decay=0.01)
        if mode == "big":
            ts.set_batch(ids, mask, pv, labels)
            ts.forward(); ts.backward()
        else:
            for h in range(2):
                sl = slice(2 * h, 2 * h + 2)
                ts.set_batch(ids[sl], mask[sl], pv[sl], labels[sl])
                ts.forward(); ts.backward()
                ts.accumulate(0.5)
            ts.use_accumulated()
            assert ts.store.grad_acc.abs().max().item() == 0.0
        grads[mode] = ts.store.grad.clone()
    c = cos(grads["big"].cpu(), grads["accum"].cpu())
    rel = ((grads["big"] - grads["accum"]).norm() / grads["big"].norm()).item()
    print(f"accumulated vs big batch: cosine {c:.6f}, rel diff {rel:.4f}")
    assert c > 0.999 and rel < 0.03


def test_lora_dropout_step_is_consistent(dev):
    """`lora_dropout` > 0 (vla-scripts/finetune.py:101,177; PEFT: base(x) + lora_B(lora_A(dropout(x)))·scaling). The oracle has
    no per-module hook for a masked adapter branch, so the step is checked through properties on the tiny model with random
    non-zero adapters: (i) p = 0 leaves the plans untouched; (ii) the mask changes the loss, and a new forward pass draws a new
    mask; (iii) with the mask held fixed, a step −ε·g along the adapter gradient lowers the loss by ε·‖g‖² to first order —
    forward (masked t), dB, dA (from the recomputed mask) and the masked share of dx are one consistent derivative."""
    from bridgelang_amd import ops
    from bridgelang_amd.training.lora import LoraAdapters
    from bridgelang_amd.training.step import TrainStep
    from bridgelang_amd.weights import allocate, tiny_dims
    dims = tiny_dims()
    w = allocate(dims, dev).fill_synthetic(seed=3)
    B, L = 3, 20
    ids, mask, labels, pv = make_batch(dims, B, L)

    def build(p):
        lora = LoraAdapters(w, r=32, dropout=p)
        lora.load_state_dict(random_adapters(lora, seed=7, b_std=0.05))
        ts = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"), weight_decay=0.0)
        ts.set_batch(ids, mask, pv, labels)
        return lora, ts
    lora0, ts0 = build(0.0)
    names0 = [op.name for op in ts0.forward_ops + ts0.backward_ops]
    assert "bl_dropout_bf16" not in names0 and "bl_dropout_grad_fix_bf16" not in names0
    base_loss = ts0.forward().item()
    del ts0, lora0
    lora, ts = build(0.25)
    names = [op.name for op in ts.vision_forward_ops + ts.forward_ops + ts.backward_ops]
    n_ad = len(lora.adapters)
    assert names.count("bl_dropout_bf16") == 2 * n_ad and names.count("bl_dropout_grad_fix_bf16") >= n_ad - 2
    l1 = ts.forward().item()
    l2 = ts.forward().item()                                       # the seed moved on: another mask
    assert l1 != base_loss and l2 != l1 and abs(l1 - base_loss) < 0.2 * base_loss
    # hold the mask: every forward below runs at device seed 5
    ts._drop_seed.fill_(4)
    loss0 = ts.forward().item()
    ts.backward()
    st = ts.store
    g = {i: (st.grad_view(f"lora.{i}.A").clone(), st.grad_view(f"lora.{i}.B").clone()) for i in range(n_ad)}
    gnorm2 = sum(a.double().pow(2).sum().item() + b.double().pow(2).sum().item() for a, b in g.values())
    A0 = [ad.A.clone() for ad in lora.adapters]
    B0 = [ad.B.clone() for ad in lora.adapters]
    deltas = []
    for target in (0.02, 0.04):
        eps = target * loss0 / gnorm2
        for i, ad in enumerate(lora.adapters):
            ad.A.copy_((A0[i].float() - eps * g[i][0].view(ad.A.shape)).to(torch.bfloat16))
            ad.B.copy_((B0[i].float() - eps * g[i][1].view(ad.B.shape)).to(torch.bfloat16))
        ops.run_all(ts._adapter_ops)
        ts._drop_seed.fill_(4)
        d = ts.forward().item() - loss0
        deltas.append((target * loss0, d))
        print(f"lora dropout directional derivative: predicted {-target * loss0:.4f}, measured {d:.4f}")
    for pred, d in deltas:
        assert abs(d + pred) <= 0.25 * pred + 2e-3, (pred, d)
    assert deltas[1][1] < deltas[0][1] < 0
    # (iv) one adapted linear in isolation (o_proj of the last layer), sharp: with a random dy the plan's dx and dA must match the
    # masked formulas — and must NOT match the un-masked ones (the whole-model derivative above is dominated by the base path)
    from oracle.synth import dropout_keep
    for i, ad in enumerate(lora.adapters):
        ad.A.copy_(A0[i]); ad.B.copy_(B0[i])
    ops.run_all(ts._adapter_ops)
    ts._drop_seed.fill_(4)
    ts.forward()                                                   # seed 5; fills ao / t of every adapted linear
    l = dims.llm_layers - 1
    packed = w.layers[l].o_w
    ad = lora.get(packed)
    Tn, D = ts.T, dims.llm_dim
    g_ = torch.Generator().manual_seed(11)
    dy = (torch.randn(Tn, D, generator=g_) * 0.5).to(torch.bfloat16)
    ts.dxb[:, :D].copy_(dy.to(dev))
    ops.run_all(ts._lin_bwd(ts.dxb, ts.ao[l], packed, ts.dao))
    x = ts.ao[l][:, :D].float().cpu()
    Wm = ops.unpack_weight(packed).float().cpu()                   # [D, D]
    A, Bm = ad.A.float().cpu()[:, :D], ad.B.float().cpu()          # [R, D], [D, R]
    keep = torch.from_numpy(dropout_keep(5, ad.index, Tn, D, 0.25))
    mt = keep.float() / 0.75
    dt = (dy.float() @ (lora.scaling * Bm)).to(torch.bfloat16).float()          # [T, R]
    dx_masked = dy.float() @ Wm + (dt @ A) * mt
    dx_plain = dy.float() @ Wm + dt @ A
    got_dx = ts.dao[:, :D].float().cpu()
    e_masked = (got_dx - dx_masked).norm() / dx_masked.norm()
    e_plain = (got_dx - dx_plain).norm() / dx_plain.norm()
    dA_masked, dA_plain = dt.t() @ (x * mt), dt.t() @ x
    got_dA = st.grad_view(f"lora.{ad.index}.A").view(ad.A.shape)[:, :D].float().cpu()
    c_masked, c_plain = cos(got_dA, dA_masked), cos(got_dA, dA_plain)
    print(f"o_proj in isolation: dx rel. error vs masked formula {e_masked:.2e} (vs un-masked {e_plain:.2e}); dA cosine {c_masked:.5f} (un-masked {c_plain:.5f})")
    assert e_masked < 1e-2 and e_plain > 3 * e_masked
    assert c_masked > 0.9995 and c_plain < 0.999
