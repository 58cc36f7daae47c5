"""debug aid: where does bl_crop_resize_bilinear_u8 differ from the host restatement?"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from bridgelang_amd.vla import eval_preprocess as EP
f32 = np.float32
H, W, out_hw, cs = 256, 256, (224, 224), 0.9
rng = np.random.default_rng(H * 131 + W)
frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
want = np.stack([EP.center_crop_and_resize(f, cs, out_hw) for f in frames])
got = EP.center_crop_and_resize_gpu(torch.from_numpy(frames).cuda(), cs, out_hw).cpu().numpy()
d = got.astype(int) - want.astype(int)
idx = np.argwhere(d != 0)
print("mismatches", len(idx), "of", d.size, "max abs", np.abs(d).max() if len(idx) else 0)
yb, ys_, xb, xs_ = [f32(v) for v in EP.sampling_constants(EP.center_crop_box(cs), (H, W), out_hw)]
k = f32(1.0) / f32(255.0)
for b, i, j, c in idx[:12]:
    ys = yb + f32(i) * ys_; xs = xb + f32(j) * xs_
    fy, fx = np.floor(ys), np.floor(xs)
    wy, wx = f32(ys - fy), f32(xs - fx)
    y0, x0 = int(fy), int(fx)
    p = lambda y, x: f32(frames[b, min(max(y, 0), H - 1), min(max(x, 0), W - 1), c]) * k
    top = p(y0, x0) * (f32(1) - wx) + p(y0, x0 + 1) * wx
    bot = p(y0 + 1, x0) * (f32(1) - wx) + p(y0 + 1, x0 + 1) * wx
    v = top * (f32(1) - wy) + bot * wy
    v255 = min(max(v, f32(0)), f32(1)) * f32(255.5)
    # the same with fused multiply-adds (what a contracting compiler would compute)
    import math
    fma = lambda a, b_, c_: f32(np.float64(a) * np.float64(b_) + np.float64(c_))
    top_f = fma(p(y0, x0), f32(1) - wx, p(y0, x0 + 1) * wx)
    bot_f = fma(p(y0 + 1, x0), f32(1) - wx, p(y0 + 1, x0 + 1) * wx)
    v_f = fma(top_f, f32(1) - wy, bot_f * wy)
    print((b, i, j, c), "got", got[b, i, j, c], "want", want[b, i, j, c], "v*255.5 =", repr(float(v255)), "with fma:", repr(float(v_f * f32(255.5))),
          "ys", repr(float(ys)), "xs", repr(float(xs)))
