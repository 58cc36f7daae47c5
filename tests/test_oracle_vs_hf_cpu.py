"""CPU: the oracle's Llama restatement vs the INSTALLED transformers LlamaForCausalLM (the reference's arithmetic lives
in transformers) on identical seeded weights: fp32 mode must agree to rounding; prefill + cached decode +
greedy ids; padding mask; shifted CE loss."""
import pytest
import torch

from oracle import restate as R

transformers = pytest.importorskip("transformers")


def build(seed=0, layers=2, hidden=128, heads=2, inter=256, vocab=320):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                      num_attention_heads=heads, num_key_value_heads=heads, rms_norm_eps=1e-6, rope_theta=10000.0,
                      max_position_embeddings=256, pad_token_id=0, attn_implementation="eager")
    torch.manual_seed(seed)
    m = LlamaForCausalLM(cfg).eval()
    for p in m.parameters():      # norm weights away from 1 so their placement matters
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    sd = {"language_model." + k: v.detach().clone() for k, v in m.state_dict().items()}
    return m, sd, cfg


def test_prefill_and_decode_match_hf_fp32():
    m, sd, cfg = build()
    B, S = 2, 19
    x = torch.randn(B, S, cfg.hidden_size, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = m(inputs_embeds=x, use_cache=True)
    p = R.Prec(False)
    logits, cache = R.llama_forward(p, sd, x, cfg.num_attention_heads, cfg.num_hidden_layers, 1e-6, 10000.0, max_pos=256)
    assert torch.allclose(logits, ref.logits, rtol=2e-4, atol=2e-4), (logits - ref.logits).abs().max()
    # one cached decode step
    nxt = ref.logits[:, -1].argmax(-1)
    emb = m.get_input_embeddings()(nxt).unsqueeze(1)
    with torch.no_grad():
        ref2 = m(inputs_embeds=emb, past_key_values=ref.past_key_values, use_cache=True)
    lg2, _ = R.llama_forward(p, sd, emb.detach(), cfg.num_attention_heads, cfg.num_hidden_layers, 1e-6, 10000.0,
                             cache=cache, max_pos=256)
    assert torch.allclose(lg2, ref2.logits, rtol=2e-4, atol=2e-4)
    assert torch.equal(lg2[:, -1].argmax(-1), ref2.logits[:, -1].argmax(-1))


def test_padding_mask_and_loss_match_hf():
    m, sd, cfg = build(seed=3)
    B, S = 3, 12
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, cfg.vocab_size, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[1, 8:] = False
    mask[2, 5:] = False
    labels = ids.clone()
    labels[~mask] = -100
    labels[:, :3] = -100
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask, labels=labels)
    x = sd["language_model.model.embed_tokens.weight"][ids]
    logits, _ = R.llama_forward(R.Prec(False), sd, x, cfg.num_attention_heads, cfg.num_hidden_layers, 1e-6, 10000.0,
                                key_mask=mask, max_pos=256)
    valid = mask.unsqueeze(-1).expand_as(logits)
    assert torch.allclose(logits[valid], ref.logits[valid], rtol=2e-4, atol=2e-4)
    loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, cfg.vocab_size), labels[:, 1:].reshape(-1),
                                             ignore_index=-100)
    assert torch.allclose(loss, ref.loss, rtol=1e-4)


def test_bf16_mode_close_to_hf_bf16():
    """bf16 rounding points: the oracle (fp32 values rounded at the reference's materialisation points) tracks an actual
    bf16 HF model within bf16 noise."""
    m, sd, cfg = build(seed=5)
    mb = m.to(torch.bfloat16)
    sdb = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    x = torch.randn(2, 16, cfg.hidden_size, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16)
    with torch.no_grad():
        ref = mb(inputs_embeds=x).logits.float()
    logits, _ = R.llama_forward(R.Prec(True), sdb, x.float(), cfg.num_attention_heads, cfg.num_hidden_layers, 1e-6,
                                10000.0, max_pos=256)
    err = (logits - ref).abs().max().item()
    assert err <= 0.05 * ref.abs().max().item(), err


# ---- vision towers vs the installed transformers' independent implementations of the same architectures ---------------
# timm 0.9.10 (the reference's dependency for the towers, modeling_prismatic.py:78-101) is not installed and the reference
# holds no tower vectors, but transformers ships its own ports: Dinov2WithRegistersModel (DINOv2 ViT with register tokens,
# LayerScale) and SiglipVisionModel (SigLIP ViT). With identical weights mapped onto timm's names, the restatement of the
# tap `get_intermediate_layers(n={depth-2})` must reproduce their hidden states (prefix tokens stripped, no final norm).
def _rand_state(model, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {k: torch.randn(v.shape, generator=g) * (0.02 if v.dim() > 1 else 0.1) + (1.0 if k.endswith("norm1.weight") or k.endswith("norm2.weight") else 0.0)
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    return sd


def test_dinov2_register_tower_matches_hf_port():
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    D, depth, heads = 64, 4, 4
    cfg = Dinov2WithRegistersConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=4, image_size=224,
                                    patch_size=14, num_register_tokens=4, layerscale_value=0.1, layer_norm_eps=1e-6,
                                    hidden_act="gelu", attn_implementation="eager")
    hf = Dinov2WithRegistersModel(cfg).eval()
    sd = _rand_state(hf, 0)
    sd["embeddings.position_embeddings"][:, 0] = 0          # timm no_embed_class: the cls token carries no position embedding
    hf.load_state_dict(sd)
    p = "vision_backbone.featurizer"
    mine = {f"{p}.cls_token": sd["embeddings.cls_token"], f"{p}.reg_token": sd["embeddings.register_tokens"],
            f"{p}.pos_embed": sd["embeddings.position_embeddings"][:, 1:],
            f"{p}.patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"],
            f"{p}.patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"]}
    for i in range(depth):
        h, b = f"encoder.layer.{i}", f"{p}.blocks.{i}"
        a = f"{h}.attention.attention"
        mine.update({f"{b}.norm1.weight": sd[f"{h}.norm1.weight"], f"{b}.norm1.bias": sd[f"{h}.norm1.bias"],
                     f"{b}.attn.qkv.weight": torch.cat([sd[f"{a}.query.weight"], sd[f"{a}.key.weight"], sd[f"{a}.value.weight"]]),
                     f"{b}.attn.qkv.bias": torch.cat([sd[f"{a}.query.bias"], sd[f"{a}.key.bias"], sd[f"{a}.value.bias"]]),
                     f"{b}.attn.proj.weight": sd[f"{h}.attention.output.dense.weight"], f"{b}.attn.proj.bias": sd[f"{h}.attention.output.dense.bias"],
                     f"{b}.ls1.scale_factor": sd[f"{h}.layer_scale1.lambda1"], f"{b}.ls2.scale_factor": sd[f"{h}.layer_scale2.lambda1"],
                     f"{b}.norm2.weight": sd[f"{h}.norm2.weight"], f"{b}.norm2.bias": sd[f"{h}.norm2.bias"],
                     f"{b}.mlp.fc1.weight": sd[f"{h}.mlp.fc1.weight"], f"{b}.mlp.fc1.bias": sd[f"{h}.mlp.fc1.bias"],
                     f"{b}.mlp.fc2.weight": sd[f"{h}.mlp.fc2.weight"], f"{b}.mlp.fc2.bias": sd[f"{h}.mlp.fc2.bias"]})
    pix = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        hs = hf(pix, output_hidden_states=True).hidden_states
    n_run = depth - 1                                                      # blocks 0 .. depth-2: the reference's tap
    got = R.vit_tower(R.Prec(False), mine, p, pix, heads, n_run)
    want = hs[n_run][:, 5:]                                                # cls + 4 registers stripped, no final norm
    assert got.shape == want.shape == (2, 256, D)
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-4 * want.abs().max().item()), (got - want).abs().max()


def test_siglip_tower_matches_hf_port():
    from transformers import SiglipVisionConfig, SiglipVisionModel
    D, depth, heads, inter = 72, 4, 4, 200                                # head_dim 18; ragged MLP width like SO400M's 4304
    cfg = SiglipVisionConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=inter,
                             image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-6, attn_implementation="eager")
    hf = SiglipVisionModel(cfg).eval()
    sd = _rand_state(hf, 2)
    hf.load_state_dict(sd)
    v = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""     # key prefix differs across versions
    p = "vision_backbone.fused_featurizer"
    mine = {f"{p}.pos_embed": sd[f"{v}embeddings.position_embedding.weight"][None],
            f"{p}.patch_embed.proj.weight": sd[f"{v}embeddings.patch_embedding.weight"],
            f"{p}.patch_embed.proj.bias": sd[f"{v}embeddings.patch_embedding.bias"]}
    for i in range(depth):
        h, b = f"{v}encoder.layers.{i}", f"{p}.blocks.{i}"
        a = f"{h}.self_attn"
        mine.update({f"{b}.norm1.weight": sd[f"{h}.layer_norm1.weight"], f"{b}.norm1.bias": sd[f"{h}.layer_norm1.bias"],
                     f"{b}.attn.qkv.weight": torch.cat([sd[f"{a}.q_proj.weight"], sd[f"{a}.k_proj.weight"], sd[f"{a}.v_proj.weight"]]),
                     f"{b}.attn.qkv.bias": torch.cat([sd[f"{a}.q_proj.bias"], sd[f"{a}.k_proj.bias"], sd[f"{a}.v_proj.bias"]]),
                     f"{b}.attn.proj.weight": sd[f"{a}.out_proj.weight"], f"{b}.attn.proj.bias": sd[f"{a}.out_proj.bias"],
                     f"{b}.norm2.weight": sd[f"{h}.layer_norm2.weight"], f"{b}.norm2.bias": sd[f"{h}.layer_norm2.bias"],
                     f"{b}.mlp.fc1.weight": sd[f"{h}.mlp.fc1.weight"], f"{b}.mlp.fc1.bias": sd[f"{h}.mlp.fc1.bias"],
                     f"{b}.mlp.fc2.weight": sd[f"{h}.mlp.fc2.weight"], f"{b}.mlp.fc2.bias": sd[f"{h}.mlp.fc2.bias"]})
    pix = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        hs = hf(pix, output_hidden_states=True).hidden_states
    n_run = depth - 1
    got = R.vit_tower(R.Prec(False), mine, p, pix, heads, n_run)
    want = hs[n_run]
    assert got.shape == want.shape == (2, 256, D)
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-4 * want.abs().max().item()), (got - want).abs().max()
