"""Measure the NOISE FLOOR between two correct fp32 summation orders at full openvla-7b size, on the CPU oracle alone.

    python tests/golden/make_noise_floor_7b.py --recipe init|decisive|margin [--order tree8|f64] [--seed 0]

Runs BASELINE configs[0] (batch 1, one 224 px frame + 32-token prompt, greedy 7 tokens; the inputs of
tests/golden/make_cfg_7b.py) through oracle/restate.py twice on one synthetic checkpoint: once in the oracle's own order
("blas": whatever the CPU BLAS does) and once with every nn.Linear's K contraction summed in a second order ("tree8": 8
K-slices, balanced tree — the order of the HIP weight-streaming kernel; "f64": the correctly rounded sum). Nothing else
differs between the two runs — same weights, same inputs, same bf16 rounding points — so every difference is what a
correct implementation may legitimately differ from the oracle by. The second run is teacher-forced on the first run's
ids so that all 7 step logits stay comparable.

Writes tests/golden/noise_floor_7b_<recipe>_b1_s<seed>_<order>.npz (data only): the second run's last-row logits of every
step (bf16 bit patterns), its argmax ids, the relative rms drift of the residual stream after every ViT block / the
projector / every decoder layer (prefill), and the summary figures tests/test_noise_floor_cpu.py re-derives and prints.
The first run must reproduce the committed cfg1_7b_<recipe>_b1_s<seed>.npz bit for bit (checked here when it exists).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def rel_rms(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a - b).double().pow(2).mean().sqrt() / a.double().pow(2).mean().sqrt()).item()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--recipe", default="decisive")
    ap.add_argument("--order", default="tree8", choices=["tree8", "f64"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--prompt-len", type=int, default=32)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-write", action="store_true", help="measure and print only (recipe calibration)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.set_flush_denormal(True)
    from bridgelang_amd import weights as W
    from oracle import restate as R, synth as S
    from test_full_size_gpu import make_inputs

    dims = W.openvla_7b_dims()
    t0 = time.time()
    sd = S.synth_state_dict(W.tensor_specs(dims, args.recipe), seed=args.wseed,
                            overlays=W.synthetic_overlays(dims, args.recipe))
    print(f"checkpoint ({args.recipe}): {time.time() - t0:.0f} s", flush=True)
    ids, pv = make_inputs(1, args.prompt_len, args.seed)
    runs = {}
    force = None
    for order in ("blas", args.order):
        t0 = time.time()
        tr: dict = {}
        with torch.no_grad():
            gen, logits = R.OracleModel.from_dims(sd, dims, order=order).generate(ids, pv, n_new=7, trace=tr, force_ids=force)
        runs[order] = (gen, logits, tr)
        if force is None:
            force = gen
        print(f"order {order}: ids {gen[0].tolist()} in {time.time() - t0:.0f} s", flush=True)
    (g0, l0, t0_), (g1, l1, t1_) = runs["blas"], runs[args.order]
    gold = ROOT / "tests" / "golden" / f"cfg1_7b_{args.recipe}_b1_s{args.seed}.npz"
    if gold.exists():
        fx = np.load(gold)
        want = torch.from_numpy(fx["logits_bf16"].astype(np.int16)).view(torch.bfloat16).float()
        same = torch.equal(want, l0)
        print(f"first run reproduces {gold.name}: {same} (ids equal {np.array_equal(fx['ids'], g0.numpy())})")
        assert same, "the oracle's own order no longer reproduces the committed fixture"
    scale = l0.abs().amax(-1)                                        # [1, 7]
    dmax = ((l1 - l0).abs().amax(-1) / scale)[0]
    biteq = (l1 == l0).float().mean(-1)[0]
    top2 = l0.topk(2, -1).values
    gap = ((top2[..., 0] - top2[..., 1]) / scale)[0]
    drift = {k: np.array([rel_rms(a, b) for a, b in zip(t0_[k], t1_[k])]) for k in ("dino", "siglip", "projector", "llm")}
    print(f"oracle[blas] vs oracle[{args.order}] on '{args.recipe}': max |dlogit|/scale per step {[f'{v:.2e}' for v in dmax.tolist()]}")
    print(f"   bit-equal logits per step {[f'{v:.3f}' for v in biteq.tolist()]}; ids equal: {torch.equal(g0, g1)}; "
          f"oracle top-2 gap/scale per step {[f'{v:.3f}' for v in gap.tolist()]}")
    for k, v in drift.items():
        print(f"   residual-stream drift {k}: first {v[0]:.2e}, last {v[-1]:.2e} ({len(v)} points)")
    if args.no_write:
        return
    out = args.out or str(ROOT / "tests" / "golden" / f"noise_floor_7b_{args.recipe}_b1_s{args.seed}_{args.order}.npz")
    np.savez_compressed(out, logits_bf16=l1.to(torch.bfloat16).view(torch.int16).numpy(), ids=g1.numpy().astype(np.int64),
                        primary_ids=g0.numpy().astype(np.int64),
                        primary_logits_bf16=l0.to(torch.bfloat16).view(torch.int16).numpy() if not gold.exists() else np.zeros(0, np.int16),
                        drift_dino=drift["dino"], drift_siglip=drift["siglip"], drift_projector=drift["projector"],
                        drift_llm=drift["llm"], max_dlogit_over_scale=dmax.numpy(), bit_equal=biteq.numpy(),
                        order=np.array(args.order), recipe=np.array(args.recipe),
                        meta=np.array([1, args.prompt_len, args.seed, args.wseed]))
    print("wrote", out)


if __name__ == "__main__":
    main()
