"""Oracle-side count of the "inside the noise" steps at BATCH 16: the 7B oracle in its second fp32
summation order ("tree8", oracle/restate.py::Prec) on the inputs of the committed batch-16 fixtures, teacher-forced on
the primary run's ids.

    python tests/golden/make_noise_floor_b16.py --recipe decisive|init [--order tree8]

The primary run (order "blas") is NOT repeated: its ids and top-32 logits are tests/golden/cfg1_7b_<recipe>_b16_s0.npz
(made by make_cfg_7b.py --batch 16 --topk 32). This script runs only oracle' on the same checkpoint and inputs and writes
tests/golden/noise_floor_7b_<recipe>_b16_s0_<order>.npz (data only):
    ids              [16, 7]      oracle' argmax per (sequence, step)
    at_primary_bf16  [16, 7, 32]  oracle' logits at the PRIMARY run's top-32 indices (bf16 bit patterns)
    topk_vals_bf16 / topk_idx     oracle' own top-32
    flips            [16, 7]      oracle' id != primary id
    dlogit_over_scale[16, 7]      max |oracle' - primary| over the primary's top-32, / the primary's logit scale
so that tests/test_cfg_7b_golden_gpu.py can compare the HIP path's flips and logit differences with what a second
CORRECT implementation shows on exactly these 112 (sequence, step) pairs.
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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--recipe", default="decisive", choices=["init", "decisive"])
    ap.add_argument("--order", default="tree8", choices=["tree8", "f64"])
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.set_flush_denormal(True)
    from bridgelang_amd import weights as W
    from oracle import restate as R, synth as S
    from test_full_size_gpu import make_inputs

    gold = np.load(ROOT / "tests" / "golden" / f"cfg1_7b_{args.recipe}_b16_s0.npz")
    batch, prompt_len, seed, wseed = (int(v) for v in gold["meta"])
    dims = W.openvla_7b_dims()
    t0 = time.time()
    sd = S.synth_state_dict(W.tensor_specs(dims, args.recipe), seed=wseed, overlays=W.synthetic_overlays(dims, args.recipe))
    print(f"checkpoint ({args.recipe}): {time.time() - t0:.0f} s", flush=True)
    ids, pv = make_inputs(batch, prompt_len, seed)
    assert np.array_equal(ids.numpy(), gold["input_ids"]), "input recipe changed since the fixture was made"
    chk = np.array([pv.float().double().sum().item(), pv.float().abs().double().sum().item()])
    assert np.allclose(chk, gold["pixel_checksum"], rtol=0, atol=0), "pixel recipe changed since the fixture was made"
    primary_ids = torch.from_numpy(gold["ids"])
    t0 = time.time()
    with torch.no_grad():
        gen, logits = R.OracleModel.from_dims(sd, dims, order=args.order).generate(ids, pv, n_new=7, force_ids=primary_ids)
    print(f"oracle[{args.order}] B={batch}: {time.time() - t0:.0f} s", flush=True)
    p_idx = torch.from_numpy(gold["topk_idx"].astype(np.int64))
    p_val = torch.from_numpy(gold["topk_vals_bf16"]).view(torch.bfloat16).float()
    at_p = logits.gather(-1, p_idx)
    scale = torch.from_numpy(gold["logit_scale"])
    d = (at_p - p_val).abs().amax(-1) / scale
    flips = gen != primary_ids
    gap = torch.from_numpy(gold["top2_gap"]) / scale
    print(f"oracle[blas] vs oracle[{args.order}] '{args.recipe}' B={batch}: max |dlogit|/scale {d.max().item():.3e}, "
          f"flips {int(flips.sum())} of {flips.numel()}; gaps at the flips {[round(v, 4) for v in gap[flips].tolist()]}")
    tk = logits.topk(32, dim=-1)
    out = ROOT / "tests" / "golden" / f"noise_floor_7b_{args.recipe}_b16_s{seed}_{args.order}.npz"
    np.savez_compressed(out, ids=gen.numpy().astype(np.int64), primary_ids=primary_ids.numpy(),
                        at_primary_bf16=at_p.to(torch.bfloat16).view(torch.int16).numpy(),
                        topk_vals_bf16=tk.values.to(torch.bfloat16).view(torch.int16).numpy(),
                        topk_idx=tk.indices.numpy().astype(np.int32), flips=flips.numpy(),
                        dlogit_over_scale=d.numpy(), order=np.array(args.order), recipe=np.array(args.recipe),
                        meta=gold["meta"])
    print("wrote", out)


if __name__ == "__main__":
    main()
