"""Experiment: is ONE staggered pipeline at batch 16 faster than TWO at batch 8 replayed side by side on two streams (the
half-batches' one-round kernels each fill half the chip and run desynchronised, so one's store tail could pass under the
other's K loop)?   python tools/exp_two_half_batches.py"""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bridgelang_amd import weights as W
from bridgelang_amd.pipeline import StaggeredDecodePipeline
dev = torch.device("cuda:0")
w = W.allocate(W.openvla_7b_dims(), dev).fill_synthetic(seed=0)


def make(B):
    p = StaggeredDecodePipeline(w, B, 32)
    g = torch.Generator().manual_seed(B)
    ids = torch.randint(3, 31000, (B, 32), generator=g); ids[:, 0] = 1
    pv = (torch.rand(B, 6, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16)
    for e in p.engines: e.set_inputs(ids.to(dev), pv.to(dev))
    p.capture()
    return p


def timed(pipes, steps=30, warm=9):
    streams = [torch.cuda.Stream(device=dev) for _ in pipes]
    def tick():
        for p, s in zip(pipes, streams):
            with torch.cuda.stream(s):
                p.step()
    for _ in range(warm): tick()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): tick()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = make(16)
t16 = timed([one])
print(f"one pipeline, batch 16: {t16:.3f} ms per step = {16 / t16 * 1e3:.1f} action-seqs/s")
del one
torch.cuda.empty_cache()
a, b = make(8), make(8)
t8 = timed([a])
print(f"one pipeline, batch 8: {t8:.3f} ms per step = {8 / t8 * 1e3:.1f} action-seqs/s")
t88 = timed([a, b])
print(f"two pipelines, batch 8 + 8 on two streams: {t88:.3f} ms per step = {16 / t88 * 1e3:.1f} action-seqs/s")
