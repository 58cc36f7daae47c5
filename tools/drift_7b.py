"""Where does the HIP path drift from the CPU oracle at full 7B size? Two halves, same script:
   python tools/drift_7b.py oracle  [--recipe decisive]   (build container, CPU): oracle intermediates → tools/_drift_RECIPE.npz
   python tools/drift_7b.py gpu     [--recipe decisive]   (GPU box): engine intermediates vs that file, per stage
Stages: fused vision features, projector output, residual stream after every Llama layer (all S rows), last-row logits.
Diagnostic tool (imports oracle/: not product code)."""
import argparse, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("side", choices=["oracle", "gpu"]); ap.add_argument("--recipe", default="decisive")
a = ap.parse_args()
from bridgelang_amd import weights as W
from test_full_size_gpu import make_inputs
dims = W.openvla_7b_dims()
ids, pv = make_inputs(1, 32, 0)
out = ROOT / "tools" / f"_drift_{a.recipe}.npz"
bits = lambda t: t.to(torch.bfloat16).view(torch.int16).numpy()
unbits = lambda x: torch.from_numpy(x.astype(np.int16)).view(torch.bfloat16).float()
if a.side == "oracle":
    from oracle import restate as R, synth as S
    torch.set_num_threads(8); torch.set_flush_denormal(True)
    sd = S.synth_state_dict(W.tensor_specs(dims, a.recipe), seed=0, overlays=W.synthetic_overlays(dims, a.recipe))
    p = R.Prec(True)
    with torch.no_grad():
        pvr = p.rb(pv.float())
        feats = R.vision_backbone(p, sd, pvr, dims.dino.heads, dims.dino.n_run, dims.siglip.heads, dims.siglip.n_run)
        proj = R.projector(p, sd, feats)
        x = R.splice(sd, ids, proj)
        tr = []
        logits, _ = R.llama_forward(p, sd, x, dims.llm_heads, dims.llm_layers, dims.rms_eps, dims.rope_theta,
                                    rows=torch.tensor([x.shape[1] - 1]), trace=tr)
    np.savez_compressed(out, feats=bits(feats), proj=bits(proj), x0=bits(x), layers=np.stack([bits(t) for t in tr]),
                        logits=bits(logits[0, 0]))
    print("wrote", out)
else:
    from bridgelang_amd import ops
    from bridgelang_amd.engine import OpenVLAEngine
    z = np.load(out)
    dev = torch.device("cuda:0")
    w = W.allocate(dims, dev).fill_synthetic(seed=0, recipe=a.recipe)
    eng = OpenVLAEngine(w, 1, 32)
    eng.set_inputs(ids.to(dev), pv.to(dev))
    rel = lambda got, ref: ((got.float().cpu() - ref).abs().max() / ref.abs().max()).item()
    rms = lambda got, ref: ((got.float().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    eq = lambda got, ref: (got.float().cpu() == ref).float().mean().item()
    def show(tag, got, ref):
        print(f"{tag:28s} max|d|/max {rel(got, ref):.2e}  rms(d)/rms {rms(got, ref):.2e}  bit-equal {eq(got, ref):.3f}", flush=True)
    eng.run_vision(); torch.cuda.synchronize()
    show("vision features", eng.feats.view(1, 256, -1), unbits(z["feats"]))
    ops.run_all(eng.projector_ops + eng.prefill_ops[:1]); torch.cuda.synchronize()
    show("projector out", eng.x[:, 1:257], unbits(z["proj"]))
    show("spliced embeddings", eng.x, unbits(z["x0"]))
    L = dims.llm_layers
    per = 7
    for l in range(L - 1):                       # the last layer runs on the last row only
        ops.run_all(eng.prefill_ops[1 + l * per:1 + (l + 1) * per]); torch.cuda.synchronize()
        if l % 4 == 3 or l < 3:
            ref = unbits(z["layers"][l])
            show(f"after layer {l}", eng.x, ref)
            show(f"   last row only", eng.x[:, -1], ref[:, -1])
    ops.run_all(eng.prefill_ops[1 + (L - 1) * per:]); torch.cuda.synchronize()
    show("last layer, last row", eng.xd, unbits(z["layers"][L - 1])[:, -1])
    show("logits (last row)", eng.logits[0, 0], unbits(z["logits"]))
