"""End-to-end GPU parity: the HIP engine (vision towers → projector → splice → Llama prefill → 6 cached decode steps →
greedy argmax) vs the CPU oracle on the SAME synthetic checkpoint (bit-identical weights via the shared integer
generator), at reduced widths the oracle finishes in seconds. Full-size (7B) behaviour is covered by size-independent
properties in test_full_size_gpu.py.

Bars. Per kernel the HIP path is bit-identical to the oracle on >= 99.9 % of elements (tests/test_ops_gpu.py at small
sizes, tests/test_per_op_full_size_gpu.py on real 7B activations). END TO END two valid fp32 summation orders already
disagree at the bf16-ulp level after a few layers, and a randomly initialised network amplifies that (tools/drift_7b.py),
so the end-to-end bounds below are max |difference| / max |reference| with the value MEASURED on MI355X + ~50 % margin,
printed by every test: vision features 8e-3 (measured 4.5e-3), projector 8e-3 (4.2e-3), first-token logits 1.2e-2
(6.5e-3), KV cache 1.2e-2 (6.9e-3 / 8.0e-3 in layers 0 / 1; one bf16 ulp of the largest element is 3.9e-3), later-step
logits 1.5e-2 (6.1e-3 … 7.9e-3). Token ids must
match the oracle wherever the oracle's own decision is not a numerical coin-flip (top-2 gap > 2 bf16 ulps of the logit);
coin-flip positions are counted and reported, never silently accepted beyond a small budget.
"""
import numpy as np
import pytest
import torch

from oracle import restate as R
from oracle import synth as S

pytestmark = pytest.mark.gpu

B, L = 3, 12
FEATS_TOL, PROJ_TOL, LOGITS0_TOL, KCACHE_TOL, LOGITS_TOL = 8e-3, 8e-3, 1.2e-2, 1.2e-2, 1.5e-2     # see the module docstring


def make_inputs(dims, batch=B, seq=L, seed=0):
    """SURVEY §8d recipe: uniform uint8 image → the two normalisations → [B,6,224,224]; ids = BOS + random + 29871."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 224, 224, 3), generator=g, dtype=torch.uint8).float() / 255.0
    img = img.permute(0, 3, 1, 2)
    mean_d, std_d = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    pv = torch.cat([(img - mean_d) / std_d, (img - 0.5) / 0.5], dim=1).to(torch.bfloat16)
    ids = torch.randint(3, 31743, (batch, seq), generator=g)
    ids[:, 0] = 1
    ids[:, -1] = 29871
    return ids, pv


@pytest.fixture(scope="module")
def setup(dev):
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    dims = W.tiny_dims()
    w = W.allocate(dims, dev).fill_synthetic(seed=7)
    sd = S.synth_state_dict(W.tensor_specs(dims), seed=7)
    oracle = R.OracleModel.from_dims(sd, dims)
    eng = OpenVLAEngine(w, B, L)
    ids, pv = make_inputs(dims)
    return dims, w, sd, oracle, eng, ids, pv


def test_synthetic_checkpoint_bit_identical(setup):
    dims, w, sd, *_ = setup
    from bridgelang_amd import weights as W
    got = w.state_dict()
    sd_pt = S.synth_state_dict(W.passthrough_specs(dims), seed=7)      # never-executed tensors, stored for export
    assert set(got) == set(sd) | set(sd_pt)
    for name, ref in {**sd, **sd_pt}.items():
        assert torch.equal(got[name].cpu().view(torch.int16), ref.view(torch.int16)), name


def rel_err(got, ref):
    return ((got.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def test_prefill_intermediates_and_logits(setup, dev):
    dims, w, sd, oracle, eng, ids, pv = setup
    from bridgelang_amd import ops
    eng.set_inputs(ids.to(dev), pv.to(dev))
    ops.run_all(eng.vision_ops + eng.projector_ops)     # x rows 1..256 are later overwritten in place by the LLM
    torch.cuda.synchronize()
    p = R.Prec(True)
    feats = R.vision_backbone(p, sd, pv.float(), dims.dino.heads, dims.dino.n_run, dims.siglip.heads, dims.siglip.n_run)
    e = rel_err(eng.feats.view(B, 256, -1), feats)
    print(f"\nvision features rel err {e:.3g}")
    assert e < FEATS_TOL, f"vision features rel err {e}"
    proj = R.projector(p, sd, feats)
    e = rel_err(eng.x[:, 1:257], proj)
    print(f"projector rel err {e:.3g}")
    assert e < PROJ_TOL, f"projector rel err {e}"
    eng.generate(ids.to(dev), pv.to(dev))
    torch.cuda.synchronize()
    logits, cache, _ = oracle.prefill(ids, pv)
    ref_last = logits[:, -1]
    got = eng.logits[0].cpu()
    scale = ref_last.abs().max().item()
    err = (got - ref_last).abs().max().item()
    print(f"\nprefill last-row logits: max abs err {err:.4g}, scale {scale:.4g}, rel {err / scale:.3g}")
    assert err <= LOGITS0_TOL * scale, f"logits err {err} vs scale {scale}"
    # KV cache parity for the first and last layer
    for l in (0, dims.llm_layers - 1):
        e = rel_err(eng.k_cache[l][:, :, :eng.S], cache.k[l])
        print(f"k cache layer {l} rel err {e:.3g}")
        assert e < KCACHE_TOL, f"k cache layer {l} rel err {e}"


def test_greedy_ids_match_oracle(setup, dev):
    dims, w, sd, oracle, eng, ids, pv = setup
    got = eng.generate(ids.to(dev), pv.to(dev)).cpu()
    ref_ids, ref_logits = oracle.generate(ids, pv, n_new=7)
    flips = 0
    for b in range(B):
        for t in range(7):
            top2 = ref_logits[b, t].topk(2).values
            gap, ulp = (top2[0] - top2[1]).item(), abs(top2[0].item()) * 2 ** -7
            if got[b, t] != ref_ids[b, t]:
                assert gap <= 2 * ulp, (f"seq {b} step {t}: id {got[b, t].item()} vs oracle {ref_ids[b, t].item()} with a "
                                        f"decisive oracle gap {gap:.4g} (ulp {ulp:.4g})")
                flips += 1
                break   # later tokens of this sequence are conditioned on a different prefix
    print(f"\ngreedy ids: {B * 7 - flips}/{B * 7} positions compared equal; {flips} sequence(s) diverged at a coin-flip")
    assert flips <= 1
    # logits of every step agree for sequences that did not diverge
    for b in range(B):
        if torch.equal(got[b], ref_ids[b]):
            err = (eng.logits[:, b].cpu() - ref_logits[b]).abs().max().item()
            print(f"sequence {b}: max |dlogit| / scale over the 7 steps {err / ref_logits[b].abs().max().item():.3g}")
            assert err <= LOGITS_TOL * ref_logits[b].abs().max().item()


def test_graph_replay_equals_eager(setup, dev):
    dims, w, sd, oracle, eng, ids, pv = setup
    eager = eng.generate(ids.to(dev), pv.to(dev)).clone()
    eager_logits = eng.logits.clone()
    eng.capture()
    eng.gen_ids.zero_(); eng.logits.zero_()
    replay = eng.generate(ids.to(dev), pv.to(dev))
    assert torch.equal(eager, replay) and torch.equal(eager_logits, eng.logits)
    eng._graph = None


def test_batch_equals_independent_calls(setup, dev):
    """Batched generation (an extension over the reference) must equal B independent batch-1 calls, bit for bit."""
    from bridgelang_amd.engine import OpenVLAEngine
    dims, w, sd, oracle, eng, ids, pv = setup
    full = eng.generate(ids.to(dev), pv.to(dev)).clone()
    full_logits = eng.logits.clone()
    one = OpenVLAEngine(w, 1, L)
    for b in range(B):
        got = one.generate(ids[b:b + 1].to(dev), pv[b:b + 1].to(dev))
        assert torch.equal(got[0], full[b]), f"sequence {b}"
        assert torch.equal(one.logits[:, 0], full_logits[:, b]), f"sequence {b} logits"


@pytest.mark.parametrize("seq", [2, 80, 150])
def test_prompt_length_extremes_vs_oracle(setup, dev, seq):
    """Edge prompt lengths: the shortest prompt predict_action can see (BOS + the 29871 it appends, S = 258) and prompts
    beyond 64 tokens (S = 336 / 406 > 320: the prefill leaves the whole-sequence attention kernel for the chunked
    online-softmax kernel + the stand-alone RoPE / KV-cache pass, decode attends over > 320 cached keys). Same bars as the
    default-length test: prefill logits within the measured 2-layer bound, ids equal wherever the oracle is decisive."""
    from bridgelang_amd.engine import OpenVLAEngine
    dims, w, sd, oracle, *_ = setup
    ids, pv = make_inputs(dims, batch=2, seq=seq, seed=100 + seq)
    eng = OpenVLAEngine(w, 2, seq)
    got = eng.generate(ids.to(dev), pv.to(dev)).cpu()
    ref_ids, ref_logits = oracle.generate(ids, pv, n_new=7)
    scale = ref_logits.abs().max().item()
    err0 = (eng.logits[0].cpu() - ref_logits[:, 0]).abs().max().item() / scale
    print(f"\nprompt length {seq} (S = {eng.S}): prefill logits err {err0:.3g}; ids {got.tolist()} vs oracle {ref_ids.tolist()}")
    assert err0 <= LOGITS0_TOL
    for b in range(2):
        for t in range(7):
            if got[b, t] != ref_ids[b, t]:
                top2 = ref_logits[b, t].topk(2).values
                assert (top2[0] - top2[1]).item() <= 2 * abs(top2[0].item()) * 2 ** -7, (seq, b, t)
                break
            err = (eng.logits[t, b].cpu() - ref_logits[b, t]).abs().max().item() / scale
            assert err <= LOGITS_TOL, (seq, b, t, err)
