"""Where do the vision towers' milliseconds go at B = 16 (openvla-7b)? Times, as HIP graphs: both towers on two streams
(as the engine runs them), each tower alone, and per op class inside each tower (GEMM shapes, LayerNorm, attention), plus
each ViT GEMM shape in isolation.   python tools/bench_vision.py [--batch 16]"""
import argparse, sys, collections
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import ops, weights as W
from bridgelang_amd.engine import OpenVLAEngine
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); a = ap.parse_args()
dev = torch.device("cuda:0")
dims = W.openvla_7b_dims()
w = W.allocate(dims, dev).fill_synthetic(seed=0)
eng = OpenVLAEngine(w, a.batch, 32)
eng.pixel_values.copy_((torch.rand(a.batch, 6, 224, 224, device=dev) * 2 - 1).to(torch.bfloat16))


def graph_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


both = graph_ms(eng.run_vision)
dino = graph_ms(lambda: ops.run_all(eng.dino_ops))
sig = graph_ms(lambda: ops.run_all(eng.siglip_ops))
fl = lambda plan: sum(op.flops for op in plan) / 1e12
print(f"B={a.batch}: both towers (2 streams) {both:.2f} ms | DINOv2 alone {dino:.2f} ms ({fl(eng.dino_ops) / dino * 1e3:.0f} TFLOP/s) | "
      f"SigLIP alone {sig:.2f} ms ({fl(eng.siglip_ops) / sig * 1e3:.0f} TFLOP/s) | sum {dino + sig:.2f} ms; "
      f"algorithmic {fl(eng.dino_ops) + fl(eng.siglip_ops):.2f} TFLOP -> {(fl(eng.dino_ops) + fl(eng.siglip_ops)) / both * 1e3:.0f} TFLOP/s overall")
for name, plan in (("DINOv2", eng.dino_ops), ("SigLIP", eng.siglip_ops)):
    # group ops by (entry point, shape signature) and time each group as its own graph
    groups = collections.OrderedDict()
    for op in plan:
        key = op.name
        if op.name == "bl_gemm_bf16":
            d = op.keep[0]
            key = f"gemm M={d.M} N={d.N} K={d.K} epi={d.epilogue}"
        groups.setdefault(key, []).append(op)
    tot = 0.0
    for key, lst in groups.items():
        ms = graph_ms(lambda lst=lst: ops.run_all(lst))
        tot += ms
        f = sum(o.flops for o in lst)
        extra = f"  {f / ms / 1e9:7.0f} TFLOP/s" if f else ""
        print(f"  {name:7s} {key:44s} x{len(lst):3d}: {ms:6.3f} ms  ({ms / len(lst) * 1e3:6.1f} us each){extra}")
    print(f"  {name:7s} sum of groups {tot:.2f} ms")
