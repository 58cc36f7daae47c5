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
