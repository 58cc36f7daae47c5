"""The noise floor between two CORRECT fp32 summation orders at full openvla-7b size.

tests/golden/noise_floor_7b_<recipe>_b1_s0_tree8.npz (written by tests/golden/make_noise_floor_7b.py in the build
container) holds the oracle run a second time with every nn.Linear's K contraction summed as 8 slices + a balanced tree
instead of the CPU BLAS order — same weights, inputs and bf16 rounding points, teacher-forced on the first run's ids. The
first run is the committed cfg1_7b_<recipe>_b1_s0.npz fixture itself. This test re-derives the oracle-vs-oracle′ figures
from the two sets of logits, prints the table DESIGN.md §4 quotes, and pins the two facts the GPU parity bounds rest on:

  * on "init" / "decisive" two correct orders differ by 2-4 % of the logit scale and share < 10 % of their logit bits —
    so the HIP path's 2.8-3.9 % against the oracle is the floor, not kernel error (tests/test_cfg_7b_golden_gpu.py
    bounds the HIP path at 1.5 x these figures);
  * on "margin" the floor is 1.4 % of the scale — half the "decisive" figure although the greedy ids stay input- and
    step-dependent — low enough that sequences whose oracle top-2 gap is >= 3 x the floor at all 7 steps can be found.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def _bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).float()


def load_noise_floor(recipe: str, order: str = "tree8"):
    """→ (primary logits [1,7,V], alt logits [1,7,V], npz). Shared with the GPU test (bounds = 1.5 x measured floor)."""
    z = np.load(GOLD / f"noise_floor_7b_{recipe}_b1_s0_{order}.npz")
    alt = _bits(z["logits_bf16"])
    prim_file = GOLD / f"cfg1_7b_{recipe}_b1_s0.npz"
    if prim_file.exists():
        p = np.load(prim_file)
        assert np.array_equal(p["ids"], z["primary_ids"]), "noise-floor fixture was not made from the committed primary run"
        prim = _bits(p["logits_bf16"])
    else:
        prim = _bits(z["primary_logits_bf16"])
    return prim, alt, z


def floor_of(recipe: str) -> float:
    prim, alt, _ = load_noise_floor(recipe)
    return float(((alt - prim).abs().amax(-1) / prim.abs().amax(-1)).max())


@pytest.mark.parametrize("recipe,lo,hi,max_biteq", [("init", 2.5e-2, 5e-2, 0.10), ("decisive", 1.5e-2, 4e-2, 0.12), ("margin", 5e-3, 2e-2, 0.4)])
def test_oracle_vs_oracle_in_a_second_summation_order(recipe, lo, hi, max_biteq):
    prim, alt, z = load_noise_floor(recipe)
    assert prim.shape == alt.shape == (1, 7, 32064) and str(z["order"]) == "tree8" and str(z["recipe"]) == recipe
    scale = prim.abs().amax(-1)
    dmax = ((alt - prim).abs().amax(-1) / scale)[0]
    biteq = (alt == prim).float().mean(-1)[0]
    assert np.allclose(dmax.numpy(), z["max_dlogit_over_scale"], rtol=1e-6) and np.allclose(biteq.numpy(), z["bit_equal"], rtol=1e-6)
    assert np.array_equal(alt.argmax(-1).numpy(), z["ids"])
    top2 = prim.topk(2, -1).values
    gap = ((top2[..., 0] - top2[..., 1]) / scale)[0]
    ids_equal = bool(np.array_equal(z["ids"], z["primary_ids"]))
    print(f"\noracle[blas] vs oracle[tree8], openvla-7b '{recipe}', cfg 1 (B = 1, 7 greedy steps):")
    print(f"  max |dlogit| / scale per step : {[f'{v:.2e}' for v in dmax.tolist()]}  (max {dmax.max():.2e})")
    print(f"  bit-equal logits per step     : {[f'{v:.3f}' for v in biteq.tolist()]}")
    print(f"  oracle top-2 gap / scale      : {[f'{v:.3f}' for v in gap.tolist()]};  ids equal: {ids_equal}")
    for k in ("dino", "siglip", "projector", "llm"):
        d = z[f"drift_{k}"]
        pts = ", ".join(f"{v:.1e}" for v in d[:: max(1, len(d) // 6)])
        print(f"  residual-stream relative rms drift, {k:9s} ({len(d):2d} taps): {pts} … last {d[-1]:.1e}")
    assert lo <= float(dmax.max()) <= hi, "the measured floor moved out of the band DESIGN.md §4 documents"
    assert float(biteq.max()) <= max_biteq
    # every step whose oracle gap exceeds 2 x the floor must keep its id between the two orders
    safe = gap > 2 * float(dmax.max())
    assert np.array_equal(z["ids"][0][safe.numpy()], z["primary_ids"][0][safe.numpy()])
    # drift grows monotonically on the whole (first tap below last tap) in every stack: accumulation, not a single bad op
    for k in ("dino", "siglip", "llm"):
        assert z[f"drift_{k}"][0] < z[f"drift_{k}"][-1]


def test_margin_checkpoint_floor_is_below_the_others():
    f = {r: floor_of(r) for r in ("init", "decisive", "margin")}
    print(f"\nnoise floor (max |dlogit| / scale over 7 steps): {f}")
    assert f["margin"] < 0.6 * f["decisive"] < f["init"]


def test_decisive_against_the_correctly_rounded_sums():
    """A third order: every Linear accumulated in float64 and rounded once ("f64" — the correctly rounded sum). The
    oracle's own BLAS order differs from it by as much as it differs from tree8 (and as the HIP path differs from the
    oracle): none of the fp32 orders is privileged, the spread between them IS the floor."""
    prim, alt, z = load_noise_floor("decisive", "f64")
    assert str(z["order"]) == "f64"
    d = ((alt - prim).abs().amax(-1) / prim.abs().amax(-1))[0]
    biteq = (alt == prim).float().mean(-1)[0]
    print(f"\noracle[blas] vs oracle[f64] on 'decisive': max |dlogit| / scale per step {[f'{v:.2e}' for v in d.tolist()]}, "
          f"bit-equal {[f'{v:.3f}' for v in biteq.tolist()]}, ids equal {bool(np.array_equal(z['ids'], z['primary_ids']))}")
    assert 1.5e-2 <= float(d.max()) <= 4.5e-2 and float(biteq.max()) <= 0.12
    assert abs(float(d.max()) - floor_of("decisive")) <= 0.5 * floor_of("decisive")


@pytest.mark.parametrize("recipe,max_flips,max_floor", [("decisive", 8, 4e-2), ("init", 14, 4e-2)])
def test_batch16_oracle_vs_oracle_flip_count(recipe, max_flips, max_floor):
    """The oracle-side yardstick for the batch-16 steps whose top-2 gap is inside the logit noise:
    tests/golden/make_noise_floor_b16.py ran oracle′ (tree8 order) on the committed batch-16 fixture's own inputs,
    teacher-forced on its ids. Re-derive flips and per-step differences from the stored logits and pin what the GPU test
    (test_cfg_7b_golden_gpu.py::test_cfg2_batch16_full_size_vs_oracle_fixture) compares the HIP path with: a second CORRECT
    implementation flips 5 (decisive) / 9 (init) of the 112 ids, every one of them at an oracle gap below 1.2 % of the scale."""
    z = np.load(GOLD / f"noise_floor_7b_{recipe}_b16_s0_tree8.npz")
    p = np.load(GOLD / f"cfg1_7b_{recipe}_b16_s0.npz")
    assert np.array_equal(z["primary_ids"], p["ids"]) and str(z["recipe"]) == recipe and str(z["order"]) == "tree8"
    assert z["ids"].shape == (16, 7)
    at_p, prim = _bits(z["at_primary_bf16"]), _bits(p["topk_vals_bf16"])
    scale = torch.from_numpy(p["logit_scale"])
    d = (at_p - prim).abs().amax(-1) / scale
    assert np.allclose(d.numpy(), z["dlogit_over_scale"], rtol=1e-6)
    flips = z["ids"] != p["ids"]
    assert np.array_equal(flips, z["flips"])
    # oracle′'s own argmax is consistent with its stored top-32 (exact bf16 ties: argmax takes the first index, topk any)
    own = _bits(z["topk_vals_bf16"])
    pos = (torch.from_numpy(z["topk_idx"].astype(np.int64)) == torch.from_numpy(z["ids"])[..., None])
    assert bool(pos.any(-1).all()) and torch.equal((own * pos).sum(-1), own[..., 0])
    gap = p["top2_gap"] / p["logit_scale"]
    # every flip of oracle′ lands on a token the ORACLE itself holds within the noise band of its best one (usually the
    # runner-up; with three or four tokens inside the band it can be the third or fourth)
    for b, t in zip(*np.nonzero(flips)):
        idx = p["topk_idx"][b, t].tolist()
        assert int(z["ids"][b, t]) in idx
        behind = float(prim[b, t, 0] - prim[b, t, idx.index(int(z["ids"][b, t]))]) / float(scale[b, t])
        assert behind <= 2.0 * float(d.max()), (b, t, behind)
    print(f"\noracle[blas] vs oracle[tree8], openvla-7b '{recipe}', B = 16: {int(flips.sum())} of {flips.size} ids flip "
          f"(oracle gaps / scale at the flips: {[round(float(g), 4) for g in gap[flips]]}); max |dlogit| / scale {float(d.max()):.3e}; "
          f"bit-equal top-32 logits {float((at_p == prim).float().mean()):.3f}")
    assert 0 < int(flips.sum()) <= max_flips and float(gap[flips].max()) <= 2.0 * float(d.max())
    assert float(d.max()) <= max_floor
