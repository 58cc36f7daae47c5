"""Full-size parity against committed oracle fixtures: the HIP engine at openvla-7b width vs oracle/restate.py run on
the build container's CPU (tests/golden/make_cfg_7b.py → tests/golden/cfg1_7b_*.npz).

  * BASELINE configs[0] — batch 1, one 224 px frame + 32-token prompt, greedy 7 tokens — on the bench checkpoint ("init")
    and on the "decisive" synthetic checkpoint (bridgelang_amd/weights.py::tensor_specs): all 32064 last-row logits of
    every step and the 7 ids. This runs the batch-1 engine (mid-M GEMM path) at full size.
  * BASELINE configs[1]'s shape — batch 16 — on both checkpoints: ids and the oracle's top-32 logits per step.

  * The WHOLE id matrix, bit-exact: 16 sequences on the "margin" checkpoint selected so that the oracle's top-2 gap is
    >= 3x the measured logit noise at all 112 (sequence, step) pairs (tests/golden/make_margin_b16.py →
    cfg2_7b_margin_b16_selected.npz): `torch.equal(got_ids, want_ids)`, no exclusions.

Bars (north_star: action-bin ids bit-exact, logits within tolerance). The only legitimate difference between the HIP
path and the oracle is the fp32 summation order inside GEMMs / softmax / norms, which flips individual bf16 roundings.
How far two CORRECT orders drift apart on each checkpoint is MEASURED, on the oracle alone, by running it in a second
summation order (tests/golden/noise_floor_7b_*.npz, tests/test_noise_floor_cpu.py): 3.92 % of the logit scale on "init",
2.78 % on "decisive", 1.40 % on "margin" at batch 1; 3.1 % on "decisive" and the figure printed below on "init" at batch 16
(tests/golden/make_noise_floor_b16.py: oracle′ on the batch-16 fixtures' own inputs, teacher-forced on the oracle's ids).
The bars are CONSTANTS of this file (the fixtures must reproduce the pinned floors, they do not set the bars):
  * logits: max |Δlogit| / max |logit| over every comparable (sequence, step) ≤ LOGIT_TOL = min(the bar every earlier
    round passed, 1.5 x the measured floor): 4.5e-2 init, 3.5e-2 decisive, 2.1e-2 margin;
  * ids on "init" / "decisive": must EQUAL the oracle's at every step whose oracle top-2 gap exceeds GAP_MIN = 2 % of the
    logit scale, and the number of steps so asserted has a stated lower bound per fixture. A sequence stops being
    comparable after its first differing id;
  * the steps below GAP_MIN at batch 16 are compared with what oracle′ does on the SAME 112 (sequence, step) pairs: every
    HIP id that differs from the oracle's must be a token the oracle holds within 2 x the noise of its best one, the number of
    HIP flips among the comparable steps may not exceed oracle′'s own flip count by more than a stated binomial margin, and
    the per-step logit difference stays within 1.5 x the batch-16 floor;
  * ids on "margin": all of them, `torch.equal`.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
from test_noise_floor_cpu import floor_of

# oracle vs oracle′ (second summation order) at batch 1, as measured in round 3 — PINNED here; the fixtures must reproduce
# them (test_pinned_floors_match_the_fixtures), so regenerating a fixture cannot move a bar
FLOOR = {"init": 3.92e-2, "decisive": 2.78e-2, "margin": 1.40e-2}
LOGIT_TOL = {"init": 4.5e-2, "decisive": 3.5e-2, "margin": 2.1e-2}   # min(the bar of rounds 1-2, 1.5 x FLOOR)
GAP_MIN = {"init": 0.02, "decisive": 0.02}                           # ids asserted wherever the oracle's gap exceeds 2 % of the scale
MIN_ID_CHECKED = {("decisive", 1): 7, ("init", 1): 6, ("decisive", 16): 70, ("init", 16): 55}   # measured 7 / 6 / 78 / 62


def test_pinned_floors_match_the_fixtures():
    for r, f in FLOOR.items():
        assert abs(floor_of(r) - f) <= 0.02 * f, (r, floor_of(r), f)


def _bf16_bits_to_f32(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).float()


_W = {}


def _weights(recipe, dev, model="7b"):
    """one checkpoint (15 GB at 7B, 26 GB at 13B) per (recipe, model), shared by the full-size tests"""
    from bridgelang_amd import weights as W
    if (recipe, model) not in _W:
        dims = W.openvla_7b_dims() if model == "7b" else W.prism_13b_dims()
        _W[(recipe, model)] = (dims, W.allocate(dims, dev).fill_synthetic(seed=0, recipe=recipe))
    return _W[(recipe, model)]


def _run(recipe, batch, dev):
    from bridgelang_amd.engine import OpenVLAEngine
    from test_full_size_gpu import make_inputs
    fx = np.load(GOLD / f"cfg1_7b_{recipe}_b{batch}_s0.npz")
    B, L, seed, wseed = [int(v) for v in fx["meta"]]
    assert (B, wseed) == (batch, 0) and str(fx["recipe"]) == recipe
    ids, pv = make_inputs(B, L, seed)
    assert np.array_equal(ids.numpy(), fx["input_ids"]), "input recipe drifted from the fixture's"
    chk = np.array([pv.float().double().sum().item(), pv.float().abs().double().sum().item()])
    assert np.allclose(chk, fx["pixel_checksum"], rtol=1e-12), "pixel recipe drifted from the fixture's"
    dims, w = _weights(recipe, dev)
    eng = OpenVLAEngine(w, B, L)
    got_ids = eng.generate(ids.to(dev), pv.to(dev)).cpu()          # [B, 7]
    got_logits = eng.logits.permute(1, 0, 2).cpu()                 # [B, 7, V]
    return fx, got_ids, got_logits


def _compare(fx, got_ids, got_logits, recipe, tag, nf=None):
    """`nf`: the oracle′ run on the same inputs (noise_floor_7b_<recipe>_b16_s0_tree8.npz) — the yardstick for the steps whose
    oracle gap is below GAP_MIN."""
    want_ids = torch.from_numpy(fx["ids"])
    gap = torch.from_numpy(fx["top2_gap"]) / torch.from_numpy(fx["logit_scale"])
    B, T = want_ids.shape
    worst, exact_frac, checked, near_ties, id_checked = 0.0, [], 0, 0, 0
    flips, per_step = [], []
    for b in range(B):
        for t in range(T):
            if t > 0 and not torch.equal(got_ids[b, :t], want_ids[b, :t]):
                break                                            # a different prefix: later steps are not comparable
            scale = float(fx["logit_scale"][b, t])
            if "logits_bf16" in fx:
                ref = _bf16_bits_to_f32(fx["logits_bf16"][b, t])
                have = got_logits[b, t]
            else:
                idx = torch.from_numpy(fx["topk_idx"][b, t]).long()
                ref = _bf16_bits_to_f32(fx["topk_vals_bf16"][b, t])
                have = got_logits[b, t][idx]
            worst = max(worst, ((have - ref).abs().max() / scale).item())
            per_step.append(((have - ref).abs().max() / scale).item())
            exact_frac.append((have == ref).float().mean().item())
            checked += 1
            if got_ids[b, t] != want_ids[b, t]:
                flips.append((b, t, float(gap[b, t])))
            if gap[b, t] > GAP_MIN[recipe]:
                id_checked += 1
                assert got_ids[b, t] == want_ids[b, t], (f"{tag} seq {b} step {t}: id {int(got_ids[b, t])} != oracle "
                                                         f"{int(want_ids[b, t])} with a decisive gap {gap[b, t]:.3f}")
            else:
                near_ties += 1
    print(f"\n{tag}: {checked} (sequence, step) logit rows compared, max |dlogit|/scale {worst:.2e} "
          f"(oracle self-noise at B = 1 {FLOOR[recipe]:.2e}, bound {LOGIT_TOL[recipe]:.2e}), bit-equal logits {np.mean(exact_frac):.3f}; "
          f"ids equal on all {id_checked} steps with oracle gap > {GAP_MIN[recipe]:.3f}, {near_ties} steps below it; "
          f"HIP flips {len(flips)} at gaps {[round(g, 4) for _, _, g in flips]}; all ids equal: {bool(torch.equal(got_ids, want_ids))}")
    assert worst <= LOGIT_TOL[recipe]
    assert id_checked >= MIN_ID_CHECKED[(recipe, B)], "the id check may not shrink"
    if nf is not None:
        # what a second CORRECT implementation does on these very (sequence, step) pairs
        assert np.array_equal(nf["primary_ids"], fx["ids"]) and str(nf["recipe"]) == recipe
        o_flips = int(nf["flips"].sum())
        floor16 = float(nf["dlogit_over_scale"].max())
        margin = int(np.ceil(2.0 * np.sqrt(o_flips + 1.0)))          # two binomial standard deviations at oracle′'s own rate
        behind = []                                                  # how far behind its best token the ORACLE holds the HIP id
        for b, t, _ in flips:
            idx = fx["topk_idx"][b, t].tolist()
            assert int(got_ids[b, t]) in idx, "a differing id must be among the oracle's top-32 at that step"
            vals = _bf16_bits_to_f32(fx["topk_vals_bf16"][b, t])
            behind.append(float(vals[0] - vals[idx.index(int(got_ids[b, t]))]) / float(fx["logit_scale"][b, t]))
        runner_up = all(x <= 2.0 * floor16 for x in behind)
        print(f"{tag}: oracle′ (tree8) on the same pairs: {o_flips} flips of {nf['flips'].size} at gaps "
              f"{[round(float(g), 4) for g in (fx['top2_gap'] / fx['logit_scale'])[nf['flips']]]}, max |dlogit|/scale {floor16:.2e}; "
              f"HIP: {len(flips)} flips of {checked} comparable (allowed {o_flips} + {margin}), every flip lands on a token "
              f"the oracle holds within 2 x that floor of its best ({[round(x, 4) for x in behind]}): {runner_up}, worst per-step |dlogit|/scale {max(per_step):.2e} (1.5 x B=16 floor = {1.5 * floor16:.2e})")
        assert runner_up, "a differing id must be a token the oracle itself holds inside the noise band of its best one"
        assert len(flips) <= o_flips + margin, "the HIP path flips more ids than a second correct implementation does"
        assert max(g for _, _, g in flips) <= 2.0 * floor16 if flips else True, "an id may flip only inside the logit noise"
        assert max(per_step) <= 1.5 * floor16
    return near_ties, checked


@pytest.mark.parametrize("recipe", ["decisive", "init"])
def test_cfg1_batch1_full_size_vs_oracle_fixture(dev, recipe):
    """BASELINE configs[0] at full 7B size through the batch-1 engine."""
    fx, got_ids, got_logits = _run(recipe, 1, dev)
    near, checked = _compare(fx, got_ids, got_logits, recipe, f"cfg1 7B {recipe} B=1")
    if recipe == "decisive":
        assert near == 0
        assert checked == 7 and torch.equal(got_ids, torch.from_numpy(fx["ids"])), \
            "the decisive checkpoint reproduces all 7 action-token ids of this sequence"
        assert ((got_ids >= 31744) & (got_ids < 32000)).all(), "greedy ids must be action tokens on this checkpoint"


@pytest.mark.parametrize("recipe", ["decisive", "init"])
def test_cfg2_batch16_full_size_vs_oracle_fixture(dev, recipe):
    """BASELINE configs[1]'s batch of 16 at full 7B size: ids + the oracle's top-32 logits per step."""
    fx, got_ids, got_logits = _run(recipe, 16, dev)
    nf = np.load(GOLD / f"noise_floor_7b_{recipe}_b16_s0_tree8.npz")
    near, checked = _compare(fx, got_ids, got_logits, recipe, f"cfg2 7B {recipe} B=16", nf=nf)
    if recipe == "decisive":
        assert checked >= 90, "most of the 112 (sequence, step) pairs must be comparable on the decisive checkpoint"


def test_margin_checkpoint_whole_id_matrix_bit_exact(dev):
    """The north-star's id bar with no exclusions: all 16 x 7 greedy action-token ids `torch.equal` to the oracle's on a
    fixture whose every oracle top-2 gap is >= 3 x the logit noise measured between two correct summation orders on
    this checkpoint; logits (the oracle's top-32 per step) within 1.5 x that noise."""
    from bridgelang_amd.engine import OpenVLAEngine
    from test_full_size_gpu import make_inputs
    fx = np.load(GOLD / "cfg2_7b_margin_b16_selected.npz")
    B, L, _, wseed = [int(v) for v in fx["meta"]]
    assert B == 16 and wseed == 0 and str(fx["recipe"]) == "margin"
    pairs = [make_inputs(1, L, int(s)) for s in fx["seq_seeds"]]
    ids, pv = torch.cat([p[0] for p in pairs]), torch.cat([p[1] for p in pairs])
    assert np.array_equal(ids.numpy(), fx["input_ids"]), "input recipe drifted from the fixture's"
    chk = np.stack([[p.float().double().sum().item(), p.float().abs().double().sum().item()] for p in pv])
    assert np.allclose(chk, fx["pixel_checksum"], rtol=1e-12), "pixel recipe drifted from the fixture's"
    gap = fx["top2_gap"] / fx["logit_scale"]
    assert gap.shape == (16, 7) and gap.min() >= 3.0 * FLOOR["margin"], "fixture gaps must be >= 3 x the measured noise floor"
    _, w = _weights("margin", dev)
    eng = OpenVLAEngine(w, B, L)
    got_ids = eng.generate(ids.to(dev), pv.to(dev)).cpu()
    got_logits = eng.logits.permute(1, 0, 2).cpu()                 # [B, 7, V]
    want_ids = torch.from_numpy(fx["ids"])
    idx = torch.from_numpy(fx["topk_idx"]).long()
    ref = _bf16_bits_to_f32(fx["topk_vals_bf16"])
    have = torch.gather(got_logits, 2, idx)
    noise = ((have - ref).abs().amax(-1) / torch.from_numpy(fx["logit_scale"])).max().item()
    print(f"\ncfg2 7B margin B=16 (selected sequences {fx['seq_seeds'].tolist()}): all ids equal: "
          f"{bool(torch.equal(got_ids, want_ids))} over {want_ids.numel()} (sequence, step) pairs, {len(set(map(tuple, want_ids.tolist())))} "
          f"distinct id rows; min oracle gap/scale {gap.min():.4f} = {gap.min() / FLOOR['margin']:.1f} x the oracle self-noise "
          f"{FLOOR['margin']:.2e}; HIP vs oracle max |dlogit|/scale {noise:.2e} (bound {LOGIT_TOL['margin']:.2e}), "
          f"bit-equal top-32 logits {(have == ref).float().mean().item():.3f}")
    assert torch.equal(got_ids, want_ids), "every action-token id of the batch-16 fixture must equal the oracle's"
    assert ((got_ids >= 31744) & (got_ids < 32000)).all()
    assert noise <= LOGIT_TOL["margin"]
    assert gap.min() >= 3.0 * noise, "the fixture's smallest gap must also clear 3 x the noise measured on THIS run"


def test_fp8_prefill_on_the_decisive_checkpoint(dev):
    """The fp8 extension (BASELINE configs[4]: W8A8 e4m3 Llama prefill projections, `OpenVLAEngine(fp8=True)`) has no
    reference counterpart; its accuracy is stated against the bf16 ORACLE on the decisive checkpoint, where logits are
    separated (on the random-init bench checkpoint only ~40 % of greedy tokens survive ANY percent-level perturbation, which
    says nothing about fp8): max |dlogit| / scale over the comparable (sequence, step) pairs of the batch-16 fixture, and
    the fraction of ids kept; an id may flip only where the oracle's top-2 gap is within twice that error. Measured: 12 % of
    the logit scale (4 x the bf16 path), 84 % of the comparable ids kept. Bounds = measured + margin."""
    from bridgelang_amd.engine import OpenVLAEngine
    from test_full_size_gpu import make_inputs
    fx = np.load(GOLD / "cfg1_7b_decisive_b16_s0.npz")
    B, L, seed, _ = [int(v) for v in fx["meta"]]
    ids, pv = make_inputs(B, L, seed)
    _, w = _weights("decisive", dev)
    eng = OpenVLAEngine(w, B, L, fp8=True)
    got_ids = eng.generate(ids.to(dev), pv.to(dev)).cpu()
    got_logits = eng.logits.permute(1, 0, 2).cpu()
    want_ids = torch.from_numpy(fx["ids"])
    gap = torch.from_numpy(fx["top2_gap"]) / torch.from_numpy(fx["logit_scale"])
    worst, rows, kept_all, n_all, kept_clear, n_clear = 0.0, 0, 0, 0, 0, 0
    errs = []
    for b in range(B):
        for t in range(7):
            if t > 0 and not torch.equal(got_ids[b, :t], want_ids[b, :t]):
                break
            idx = torch.from_numpy(fx["topk_idx"][b, t]).long()
            ref = _bf16_bits_to_f32(fx["topk_vals_bf16"][b, t])
            e = ((got_logits[b, t][idx] - ref).abs().max() / float(fx["logit_scale"][b, t])).item()
            errs.append((e, b, t))
            worst, rows = max(worst, e), rows + 1
            n_all += 1
            kept_all += int(got_ids[b, t] == want_ids[b, t])
    flips = [(b, t, float(gap[b, t])) for e, b, t in errs if got_ids[b, t] != want_ids[b, t]]
    print(f"\nfp8 prefill vs bf16 oracle, decisive B=16: {rows} comparable rows, max |dlogit|/scale {worst:.3f} (bf16 path: 0.029); "
          f"ids kept {kept_all}/{n_all} = {kept_all / n_all:.2f}; oracle gap/scale at the flipped steps {[round(g, 3) for _, _, g in flips]}")
    assert worst <= 0.16 and rows >= 30            # measured 0.123: 4 x the bf16 path's floor
    assert all(g <= 2 * worst for _, _, g in flips), "an id may only flip where two logits within the fp8 error of each other can swap"
    assert kept_all >= 0.7 * n_all                 # measured 52 / 62 = 0.84
