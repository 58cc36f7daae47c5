"""The two-deep batch pipeline (stage 1 of batch i+1 ‖ stage 2 of batch i on two streams, captured as HIP graphs) must
return exactly the ids the plain engine returns for every submitted batch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipeline_equals_engine(dev):
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    from bridgelang_amd.pipeline import TwoStagePipeline
    from test_engine_gpu import make_inputs
    dims = W.tiny_dims()
    w = W.allocate(dims, dev).fill_synthetic(seed=3)
    B, L = 2, 9
    eng = OpenVLAEngine(w, B, L)
    batches = [make_inputs(dims, B, L, seed=s) for s in range(5)]
    want = [eng.generate(i.to(dev), p.to(dev)).clone() for i, p in batches]
    for graphs in (False, True):
        pipe = TwoStagePipeline(w, B, L)
        if graphs:
            for e in pipe.engines:
                e.set_inputs(batches[0][0].to(dev), batches[0][1].to(dev))
            pipe.capture()
        got = []
        for n, (i, p) in enumerate(batches):
            out = pipe.step(i.to(dev), p.to(dev)).clone()
            if n > 0:
                got.append(out)
        got.append(pipe.flush().clone())
        for n, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), f"graphs={graphs} batch {n}"
