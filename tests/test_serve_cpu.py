"""/act serving shell (bridgelang_amd/serve.py) on CPU with a stand-in model: wire format, the reference's request /
response contract (deploy.py:91-123) incl. double-encoded payloads and the "error" answer, prompt templates, and request
coalescing into batches."""
import json
import threading

import numpy as np
import pytest
import torch

from bridgelang_amd import serve


class FakeVLA:
    """predict_action = a deterministic function of (last prompt id, mean pixel): lets the test tell requests apart."""
    norm_stats = {"bridge_orig": {"action": {"q01": [0.0] * 7, "q99": [1.0] * 7}}}

    def __init__(self):
        self.calls = []

    def predict_action(self, input_ids=None, pixel_values=None, unnorm_key=None, do_sample=False):
        assert do_sample is False
        if unnorm_key not in (None, "bridge_orig"):
            raise AssertionError("unknown unnorm_key")
        self.calls.append(tuple(input_ids.shape))
        base = input_ids[:, -1].double().numpy()[:, None] + pixel_values.double().mean(dim=(1, 2, 3)).numpy()[:, None]
        out = base + np.arange(7)[None, :]
        return out[0] if out.shape[0] == 1 else out


class FakeProcessor:
    def __call__(self, prompt, image):
        ids = torch.tensor([[1] + [3 + (ord(c) % 50) for c in prompt][:20] + [len(prompt)]])
        px = torch.from_numpy(np.asarray(image, dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
        return {"input_ids": ids, "pixel_values": px}


def expected(proc, prompt_instruction, image):
    prompt = serve.get_openvla_prompt(prompt_instruction, "openvla/openvla-7b")
    x = proc(prompt, image)
    return x["input_ids"][0, -1].item() + x["pixel_values"].double().mean().item() + np.arange(7)


@pytest.fixture
def server():
    s = serve.OpenVLAServer(FakeVLA(), FakeProcessor(), max_batch=4, max_wait_ms=50)
    yield s
    s.close()


def test_wire_format_round_trip():
    for a in (np.arange(12, dtype=np.uint8).reshape(2, 2, 3), np.linspace(-1, 1, 7), np.float32(2.5), np.zeros((0, 3), np.int64)):
        d = serve.encode_ndarray(a)
        assert set(d) == {"__numpy__", "dtype", "shape"} and d["shape"] == list(np.asarray(a).shape)
        back = serve.loads(serve.dumps({"x": a}))["x"]
        assert np.array_equal(back, a) and np.asarray(back).dtype == np.asarray(a).dtype
    # the format is self-describing JSON: little-endian descr + base64 of the C-order bytes
    d = serve.encode_ndarray(np.array([1, 2, 3], dtype="<i4"))
    assert d["dtype"] == "<i4" and d["__numpy__"] == "AQAAAAIAAAADAAAA"


def test_prompt_templates():
    assert serve.get_openvla_prompt("Pick UP the Cup", "openvla/openvla-7b") == \
        "In: What action should the robot take to pick up the cup?\nOut:"
    p = serve.get_openvla_prompt("Pick UP the Cup", "openvla/openvla-v01-7b")
    assert p.startswith(serve.SYSTEM_PROMPT) and p.endswith("USER: What action should the robot take to pick up the cup? ASSISTANT:")


def test_act_plain_double_encoded_and_error(server):
    from starlette.testclient import TestClient
    client = TestClient(server.build_app())
    img = np.random.default_rng(0).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    want = expected(server.processor, "grasp the snack bag", img)
    # json-numpy client: ndarray fields in the body, ndarray in the answer
    r = client.post("/act", content=serve.dumps({"image": img, "instruction": "grasp the snack bag"}),
                    headers={"content-type": "application/json"})
    assert r.status_code == 200
    got = serve.loads(r.text)
    assert isinstance(got, np.ndarray) and got.dtype == np.float64 and np.allclose(got, want)
    # double-encoded: {"encoded": "<json>"} → a JSON string holding the encoded action
    body = {"encoded": serve.dumps({"image": img, "instruction": "grasp the snack bag", "unnorm_key": "bridge_orig"})}
    r = client.post("/act", json=body)
    inner = r.json()
    assert isinstance(inner, str) and np.allclose(serve.loads(inner), want)
    # malformed requests are answered with the string "error" (deploy.py:112-121), not an HTTP failure
    for bad in ({"instruction": "no image"}, {"encoded": "{}", "extra": 1},
                {"image": serve.encode_ndarray(img), "instruction": "x", "unnorm_key": "nope"}):
        r = client.post("/act", json=bad)
        assert r.status_code == 200 and r.json() == "error"


def test_concurrent_requests_are_coalesced(server):
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, (16, 16, 3), dtype=np.uint8) for _ in range(6)]
    out = [None] * 6

    def worker(i):
        out[i] = server.predict_action({"image": serve.encode_ndarray(imgs[i]), "instruction": "stack the blocks"})

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(6):
        assert np.allclose(serve.decode_tree(out[i]), expected(server.processor, "stack the blocks", imgs[i]))
    assert sum(server.batch_sizes) == 6 and max(server.batch_sizes) >= 2 and max(server.batch_sizes) <= 4
    # a different prompt length never shares a batch
    a = server.predict_action({"image": serve.encode_ndarray(imgs[0]), "instruction": "a"})
    assert np.allclose(serve.decode_tree(a), expected(server.processor, "a", imgs[0]))


def test_norm_stats_loaded_from_run_dir(tmp_path):
    (tmp_path / "dataset_statistics.json").write_text(json.dumps({"my_robot": {"action": {"q01": [0] * 7, "q99": [1] * 7}}}))
    vla = FakeVLA()
    s = serve.OpenVLAServer(vla, FakeProcessor(), openvla_path=tmp_path)
    assert list(vla.norm_stats) == ["my_robot"]
    s.close()
