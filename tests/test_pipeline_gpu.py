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


def _check_ids(got, want, want_logits, tag):
    """The merged decode iteration reproduces the per-batch kernels' arithmetic (pipeline._plan_merged), so the ids of
    every sequence must EQUAL the plain engine's — no near-tie allowance."""
    assert torch.equal(got, want), f"{tag}: ids differ in {(got != want).any(dim=1).sum().item()} of {got.shape[0]} sequences"
    return 0


@pytest.mark.parametrize("B,split", [(2, False), (6, False), (3, True)])
def test_staggered_pipeline_matches_engine(dev, B, split):
    """StaggeredDecodePipeline: every submitted batch comes out n_new-1 steps later with exactly the plain engine's ids
    and logits (the merged decode rows run through kernels with the per-batch kernels' summation order), for eager and
    graph execution, incl. the drain."""
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    from bridgelang_amd.pipeline import StaggeredDecodePipeline
    from test_engine_gpu import make_inputs
    dims = W.tiny_dims()
    w = W.allocate(dims, dev).fill_synthetic(seed=3)
    L, n = 9, 11
    eng = OpenVLAEngine(w, B, L)
    batches = [make_inputs(dims, B, L, seed=s) for s in range(n)]
    want, want_lg = [], []
    for i, p in batches:
        want.append(eng.generate(i.to(dev), p.to(dev)).clone().cpu())
        want_lg.append(eng.logits.clone().cpu())
    total_diff = 0
    for graphs in (False, True):
        pipe = StaggeredDecodePipeline(w, B, L, split_vision=split)
        if graphs:
            for e in pipe.engines:
                e.set_inputs(batches[0][0].to(dev), batches[0][1].to(dev))
            pipe.capture()
        got = []
        for k, (i, p) in enumerate(batches):
            out = pipe.step(i.to(dev), p.to(dev)).clone()
            if k >= pipe.slots - 1:
                got.append(out.cpu())
            for g in range(1, pipe.n_new):       # this step ran decode iteration g of batch j: logits bit-equal
                j = k - g - pipe.lag
                if j >= 0:
                    assert torch.equal(pipe.logits[(g - 1) * B:g * B].cpu(), want_lg[j][g]), \
                        f"graphs={graphs} step {k}: logits of batch {j}, decode iteration {g} differ from the engine's"
        got += [o.cpu() for o in pipe.flush()]
        assert len(got) == n
        for k in range(n):
            total_diff += _check_ids(got[k], want[k], want_lg[k], f"graphs={graphs} batch {k}")
    assert total_diff == 0


@pytest.mark.parametrize("split", [False, True])
def test_staggered_pipeline_short_run_drains(dev, split):
    """Fewer submissions than pipeline slots: flush() returns exactly the submitted batches."""
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    from bridgelang_amd.pipeline import StaggeredDecodePipeline
    from test_engine_gpu import make_inputs
    dims = W.tiny_dims()
    w = W.allocate(dims, dev).fill_synthetic(seed=3)
    B, L = 2, 9
    eng = OpenVLAEngine(w, B, L)
    batches = [make_inputs(dims, B, L, seed=s) for s in range(3)]
    pipe = StaggeredDecodePipeline(w, B, L, split_vision=split)
    for i, p in batches:
        pipe.step(i.to(dev), p.to(dev))
    outs = pipe.flush()
    assert len(outs) == 3
    for (i, p), o in zip(batches, outs):
        want = eng.generate(i.to(dev), p.to(dev)).clone().cpu()
        _check_ids(o.cpu(), want, eng.logits.clone().cpu(), "short run")


@pytest.mark.parametrize("n_new", [2, 4])
def test_staggered_pipeline_other_depths(dev, n_new):
    """The slot rotation is generic in n_new (slots = n_new, n_new - 1 merged decode groups)."""
    from bridgelang_amd import weights as W
    from bridgelang_amd.engine import OpenVLAEngine
    from bridgelang_amd.pipeline import StaggeredDecodePipeline
    from test_engine_gpu import make_inputs
    dims = W.tiny_dims()
    w = W.allocate(dims, dev).fill_synthetic(seed=3)
    B, L, n = 2, 9, 7
    eng = OpenVLAEngine(w, B, L, n_new=n_new)
    batches = [make_inputs(dims, B, L, seed=s) for s in range(n)]
    want, want_lg = [], []
    for i, p in batches:
        want.append(eng.generate(i.to(dev), p.to(dev)).clone().cpu())
        want_lg.append(eng.logits.clone().cpu())
    pipe = StaggeredDecodePipeline(w, B, L, n_new=n_new)
    assert pipe.slots == n_new and len(pipe.engines) == n_new
    got = []
    for k, (i, p) in enumerate(batches):
        out = pipe.step(i.to(dev), p.to(dev)).clone()
        if k >= pipe.slots - 1:
            got.append(out.cpu())
    got += [o.cpu() for o in pipe.flush()]
    assert len(got) == n and all(g.shape == (B, n_new) for g in got)
    for k in range(n):
        _check_ids(got[k], want[k], want_lg[k], f"n_new={n_new} batch {k}")
