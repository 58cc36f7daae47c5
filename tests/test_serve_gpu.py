"""/act server end to end on the HIP path (reduced-width model, character-level stand-in tokenizer — no tokenizer files
exist offline): HTTP answers equal direct `predict_action` calls, concurrent clients share GPU batches."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class CharTokenizer:
    """Deterministic stand-in for the Llama tokenizer: BOS + one id per character (fixed-length prompts batch together)."""

    def __call__(self, text, return_tensors="pt", **_):
        texts = [text] if isinstance(text, str) else list(text)
        ids = torch.tensor([[1] + [3 + (ord(c) * 131) % 31000 for c in t] for t in texts])
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}


def test_act_server_matches_predict_action(dev):
    from starlette.testclient import TestClient
    from PIL import Image
    from bridgelang_amd import serve, weights as W
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import OpenVLAForActionPrediction
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticProcessor
    stats = {"bridge_orig": {"action": {"q01": [-0.5] * 7, "q99": [0.7] * 7, "mask": [True] * 6 + [False]}}}
    vla = OpenVLAForActionPrediction(OpenVLAConfig(norm_stats=stats), device=dev, dims=W.tiny_dims()).init_synthetic(seed=11)
    proc = PrismaticProcessor(tokenizer=CharTokenizer())
    server = serve.OpenVLAServer(vla, proc, max_batch=4, max_wait_ms=100)
    client = TestClient(server.build_app())
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(5)]
    instr = "grasp the snack bag"

    def direct(img):
        x = proc(serve.get_openvla_prompt(instr, "openvla/openvla-7b"), Image.fromarray(img).convert("RGB"))
        return vla.predict_action(input_ids=x["input_ids"].to(dev), pixel_values=x["pixel_values"].to(dev, torch.bfloat16),
                                  unnorm_key="bridge_orig", do_sample=False)

    want = [direct(i) for i in imgs]
    r = client.post("/act", content=serve.dumps({"image": imgs[0], "instruction": instr}), headers={"content-type": "application/json"})
    got0 = serve.loads(r.text)
    assert got0.shape == (7,) and np.array_equal(got0, want[0])
    # concurrent clients: coalesced into GPU batches, each gets its own sequence's action
    out = [None] * 5

    def worker(i):
        out[i] = serve.decode_tree(server.predict_action({"image": serve.encode_ndarray(imgs[i]), "instruction": instr,
                                                          "unnorm_key": "bridge_orig"}))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(5)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(5):
        assert np.array_equal(out[i], want[i]), f"request {i}"
    assert max(server.batch_sizes) >= 2
    assert client.post("/act", json={"instruction": "no image"}).json() == "error"
    server.close()


def test_act_server_throughput_mode(dev):
    """pipeline_batch=B: requests ride the StaggeredDecodePipeline (a batch per tick, merged decode of the older batches,
    immediate drain when the queue runs dry). Every request is answered with its own sequence's action; answers equal the
    plain engine's except at near ties of the synthetic model (other decode GEMM kernel, see test_pipeline_gpu.py)."""
    from PIL import Image
    from bridgelang_amd import serve, weights as W
    from bridgelang_amd.extern.hf.configuration_prismatic import OpenVLAConfig
    from bridgelang_amd.extern.hf.modeling_prismatic import OpenVLAForActionPrediction
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticProcessor
    stats = {"bridge_orig": {"action": {"q01": [-0.5] * 7, "q99": [0.7] * 7, "mask": [True] * 6 + [False]}}}
    vla = OpenVLAForActionPrediction(OpenVLAConfig(norm_stats=stats), device=dev, dims=W.tiny_dims()).init_synthetic(seed=11)
    proc = PrismaticProcessor(tokenizer=CharTokenizer())
    instr = "stack the blocks"
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(23)]

    def direct(img):
        x = proc(serve.get_openvla_prompt(instr, "openvla/openvla-7b"), Image.fromarray(img).convert("RGB"))
        return vla.predict_action(input_ids=x["input_ids"].to(dev), pixel_values=x["pixel_values"].to(dev, torch.bfloat16),
                                  unnorm_key="bridge_orig", do_sample=False)

    want = [direct(i) for i in imgs]
    server = serve.OpenVLAServer(vla, proc, pipeline_batch=2, max_wait_ms=20)
    out = [None] * len(imgs)

    def worker(i):
        out[i] = serve.decode_tree(server.predict_action({"image": serve.encode_ndarray(imgs[i]), "instruction": instr,
                                                          "unnorm_key": "bridge_orig"}))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(imgs))]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert all(isinstance(o, np.ndarray) and o.shape == (7,) and np.isfinite(o).all() for o in out), out
    same = sum(bool(np.array_equal(o, w)) for o, w in zip(out, want))
    print(f"\nthroughput-mode server: {same}/{len(imgs)} answers identical to the plain engine; GPU batches {server.batch_sizes}")
    assert same >= int(0.6 * len(imgs))
    assert sum(server.batch_sizes) == len(imgs) and len(server.batch_sizes) >= 12
    # a lone request afterwards: one tick + drain
    lone = serve.decode_tree(server.predict_action({"image": serve.encode_ndarray(imgs[0]), "instruction": instr}))
    assert lone.shape == (7,)
    # a bad unnorm_key fails that request only
    assert server.predict_action({"image": serve.encode_ndarray(imgs[1]), "instruction": instr, "unnorm_key": "nope"}) == "error"
    ok = serve.decode_tree(server.predict_action({"image": serve.encode_ndarray(imgs[1]), "instruction": instr}))
    assert ok.shape == (7,)
    server.close()
