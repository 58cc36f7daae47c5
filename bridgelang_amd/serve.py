"""`/act` REST server over the HIP path — the serving shell of vla-scripts/deploy.py.

Contract kept from the reference (deploy.py:66-123): POST /act with
    {"image": ndarray[H, W, 3] uint8, "instruction": str, "unnorm_key": Optional[str]}      → the action (ndarray[7])
or the "double-encoded" form {"encoded": "<json text of the same dict>"} → a JSON *string* holding the encoded action
(for clients without json-numpy); any failure is logged and answered with the string "error" (deploy.py:112-121).
Prompt templates as `get_openvla_prompt` (deploy.py:56-60).

ndarrays travel in the json-numpy wire format {"__numpy__": base64(bytes), "dtype": descr, "shape": [...]}. The
json-numpy package is not installed in this image, so the codec below is a restatement from its published format
(parity unpinned: no fixture of it exists in the reference).

MI355X-first difference: requests are COALESCED. The reference serves one request per forward (batch 1, ≈ 33 ms here);
one GPU pass over 16 sequences costs 83 ms, so concurrent clients are batched: a worker thread collects up to
`max_batch` requests that share prompt length and `unnorm_key` (waiting at most `max_wait_ms` for company) and runs ONE
batched `predict_action`; per-sample results equal independent batch-1 calls (tests/test_engine_gpu.py).

Throughput mode (`pipeline_batch=B`): under sustained load the worker drives `StaggeredDecodePipeline` instead — every
tick submits one batch of B requests (vision + prefill) while the six older batches advance one decode iteration in a
single merged pass over the weights (pipeline.py; the mode `bench.py` measures, 241 vs 193 action-seqs/s at B = 16). A
request is answered 7 ticks after its batch was submitted; when the queue runs dry the pipeline is drained at once
(`flush`), so a lone request still returns after one prefill + six plain decode steps. Partial batches are padded with
copies of their last request.
"""
from __future__ import annotations

import base64
from collections import OrderedDict
import json
import logging
import queue
import threading
import traceback
from concurrent.futures import Future
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

SYSTEM_PROMPT = (
    "A chat between a curious user and an artificial intelligence assistant. "
    "The assistant gives helpful, detailed, and polite answers to the user's questions."
)


def get_openvla_prompt(instruction: str, openvla_path: Union[str, Path]) -> str:
    """deploy.py:56-60: v01 checkpoints use the Vicuna chat template, everything else the pure `In:/Out:` one."""
    if "v01" in str(openvla_path):
        return f"{SYSTEM_PROMPT} USER: What action should the robot take to {instruction.lower()}? ASSISTANT:"
    return f"In: What action should the robot take to {instruction.lower()}?\nOut:"


# ---- json-numpy wire format ------------------------------------------------------------------------------------------
def encode_ndarray(a: Union[np.ndarray, np.generic]) -> Dict[str, Any]:
    a = np.asarray(a)
    descr = np.lib.format.dtype_to_descr(a.dtype)
    return {"__numpy__": base64.b64encode(np.ascontiguousarray(a).tobytes()).decode("ascii"), "dtype": descr,
            "shape": list(a.shape)}


def _default(o: Any) -> Any:
    if isinstance(o, (np.ndarray, np.generic)):
        return encode_ndarray(o)
    raise TypeError(f"Object of type {type(o).__name__} is not JSON serializable")


def _hook(d: Dict[str, Any]) -> Any:
    if "__numpy__" in d:
        dt = np.lib.format.descr_to_dtype(d["dtype"])
        flat = np.frombuffer(base64.b64decode(d["__numpy__"]), dtype=dt)
        shape = tuple(d.get("shape", ()))
        return flat.reshape(shape).copy() if shape else flat[0]
    return d


def dumps(obj: Any) -> str:
    return json.dumps(obj, default=_default)


def loads(text: Union[str, bytes]) -> Any:
    return json.loads(text, object_hook=_hook)


def decode_tree(obj: Any) -> Any:
    """Apply the ndarray hook to an already-parsed JSON tree (FastAPI hands the handler plain dicts)."""
    if isinstance(obj, dict):
        return _hook({k: decode_tree(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [decode_tree(v) for v in obj]
    return obj


# ---- server ----------------------------------------------------------------------------------------------------------
class _Request:
    __slots__ = ("input_ids", "pixel_values", "unnorm_key", "future")

    def __init__(self, input_ids, pixel_values, unnorm_key):
        self.input_ids, self.pixel_values, self.unnorm_key = input_ids, pixel_values, unnorm_key
        self.future: Future = Future()


class OpenVLAServer:
    """`vla`: OpenVLAForActionPrediction (HIP) — anything with `predict_action(input_ids=, pixel_values=, unnorm_key=,
    do_sample=False) -> ndarray [B, 7] (or [7] at B = 1)`; `processor(prompt, PIL image) -> {input_ids, pixel_values}`."""

    def __init__(self, vla: Any, processor: Any, openvla_path: Union[str, Path] = "openvla/openvla-7b",
                 max_batch: int = 16, max_wait_ms: float = 2.0, norm_stats_path: Optional[Union[str, Path]] = None,
                 pipeline_batch: Optional[int] = None, max_pipelines: int = 2):
        self.vla, self.processor, self.openvla_path = vla, processor, str(openvla_path)
        self.max_batch, self.max_wait = int(max_batch), float(max_wait_ms) * 1e-3
        self.pipeline_batch = pipeline_batch
        if pipeline_batch:
            self.max_batch = int(pipeline_batch)
        # prompt length → (StaggeredDecodePipeline, {tick: requests}); least recently used first. A pipeline is 7 engines
        # (KV caches + activations, ≈ 3 GB each at B = 16 on 7B) + 7 graphs, and instruction lengths vary per request, so
        # only `max_pipelines` are kept; the plain path's engines are bounded the same way on the model (engine LRU).
        self._pipes: "OrderedDict[int, Any]" = OrderedDict()
        self.max_pipelines = max(1, int(max_pipelines))
        stats = Path(norm_stats_path) if norm_stats_path else Path(self.openvla_path) / "dataset_statistics.json"
        if stats.is_file():                       # fine-tuned run directory (deploy.py:86-89)
            self.vla.norm_stats = json.loads(stats.read_text())
        self.batch_sizes: List[int] = []          # sizes of the GPU batches run so far (observability / tests)
        self._q: "queue.Queue[Optional[_Request]]" = queue.Queue()
        self._held: Optional[_Request] = None     # a request that did not fit the previous batch
        self._worker = threading.Thread(target=self._serve_loop, name="openvla-batcher", daemon=True)
        self._worker.start()

    # -- request side (any thread) --
    def _submit(self, payload: Dict[str, Any]) -> np.ndarray:
        from PIL import Image
        image, instruction = payload["image"], payload["instruction"]
        unnorm_key = payload.get("unnorm_key", None)
        prompt = get_openvla_prompt(instruction, self.openvla_path)
        inputs = self.processor(prompt, Image.fromarray(np.asarray(image, dtype=np.uint8)).convert("RGB"))
        req = _Request(inputs["input_ids"], inputs["pixel_values"], unnorm_key)
        self._q.put(req)
        return req.future.result()

    def predict_action(self, payload: Dict[str, Any]) -> Any:
        """The /act handler body (deploy.py:91-121), returning the JSON-ready response object."""
        try:
            double_encode = "encoded" in payload
            if double_encode:
                assert len(payload.keys()) == 1, "Only uses encoded payload!"
                payload = loads(payload["encoded"])
            else:
                payload = decode_tree(payload)
            action = self._submit(payload)
            return dumps(action) if double_encode else encode_ndarray(action)
        except Exception:   # noqa: BLE001 — the reference answers every failure with "error"
            logging.error(traceback.format_exc())
            logging.warning(
                "Your request threw an error; make sure your request complies with the expected format:\n"
                "{'image': np.ndarray, 'instruction': str}\n"
                "You can optionally an `unnorm_key: str` to specific the dataset statistics you want to use for "
                "de-normalizing the output actions.")
            return "error"

    # -- GPU side (one thread owns the model) --
    def _take_batch(self) -> Optional[List[_Request]]:
        import time
        first = self._held if self._held is not None else self._q.get()
        self._held = None
        if first is None:
            return None
        batch, key = [first], (tuple(first.input_ids.shape), first.unnorm_key)
        deadline = time.monotonic() + self.max_wait
        while len(batch) < self.max_batch:
            try:
                nxt = self._q.get(timeout=max(0.0, deadline - time.monotonic()))
            except queue.Empty:
                break
            if nxt is None:
                self._q.put(None)
                break
            if (tuple(nxt.input_ids.shape), nxt.unnorm_key) != key:
                self._held = nxt               # different prompt length / statistics: heads the next batch
                break
            batch.append(nxt)
        return batch

    # -- throughput mode: one pipeline per prompt length --
    def _pipe_for(self, L: int):
        from .pipeline import StaggeredDecodePipeline
        if L in self._pipes:
            self._pipes.move_to_end(L)
            return self._pipes[L]
        while len(self._pipes) >= self.max_pipelines:       # evict the least recently used pipeline (always drained:
            old_len, (old, infl) = next(iter(self._pipes.items()))      # _serve_pipelined drains before switching lengths)
            assert not infl, "evicting a pipeline with batches in flight"
            del self._pipes[old_len]
            del old
            torch.cuda.empty_cache()
        pipe = StaggeredDecodePipeline(self.vla.weights, self.pipeline_batch, L)
        pipe.capture()
        self._pipes[L] = (pipe, {})
        return self._pipes[L]

    def _resolve(self, reqs: List[_Request], token_ids: torch.Tensor) -> None:
        ids = token_ids.cpu().numpy()
        for i, r in enumerate(reqs):
            try:
                r.future.set_result(np.asarray(self.vla.actions_from_token_ids(ids[i:i + 1], r.unnorm_key)).reshape(-1))
            except Exception as e:   # noqa: BLE001 — e.g. an unknown unnorm_key: that request alone fails
                r.future.set_exception(e)

    def _drain(self) -> None:
        for pipe, inflight in self._pipes.values():
            if inflight:
                outs = pipe.flush(ticks=set(inflight))    # oldest first; None for slots answered earlier
                for t, out in zip(range(pipe._tick - len(outs), pipe._tick), outs):
                    if out is not None:
                        self._resolve(inflight.pop(t), out)
                inflight.clear()

    def _serve_pipelined(self) -> None:
        dev = self.vla.device
        while True:
            busy = any(inflight for _, inflight in self._pipes.values())
            if busy and self._held is None and self._q.empty():
                self._drain()                             # nothing waiting: finish what is in flight right away
                continue
            batch = self._take_batch()
            if batch is None:
                self._drain()
                return
            try:
                if self.vla.get_action_dim(batch[0].unnorm_key) != 7:
                    # the pipelines decode 7 tokens; other action dimensions go through the plain engine
                    self._run_plain(batch)
                    continue
                ids = self.vla.with_empty_token(torch.cat([r.input_ids for r in batch], dim=0).to(dev))
                pv = torch.cat([r.pixel_values for r in batch], dim=0).to(dev, torch.bfloat16)
                pad = self.pipeline_batch - len(batch)
                if pad:
                    ids = torch.cat([ids, ids[-1:].expand(pad, -1)], dim=0)
                    pv = torch.cat([pv, pv[-1:].expand(pad, -1, -1, -1)], dim=0)
                for L2, (other, infl) in self._pipes.items():      # one pipeline at a time keeps the tick bookkeeping simple
                    if L2 != ids.shape[1] and infl:
                        self._drain()
                pipe, inflight = self._pipe_for(ids.shape[1])
                tick = pipe._tick
                out = pipe.step(ids, pv)
                inflight[tick] = batch
                self.batch_sizes.append(len(batch))
                done = tick - (pipe.slots - 1)
                if done in inflight:
                    self._resolve(inflight.pop(done), out.clone())
            except Exception as e:   # noqa: BLE001
                for r in batch:
                    if not r.future.done():
                        r.future.set_exception(e)

    def _serve_loop(self) -> None:
        if self.pipeline_batch:
            return self._serve_pipelined()
        while True:
            batch = self._take_batch()
            if batch is None:
                return
            self._run_plain(batch)

    def _run_plain(self, batch: List[_Request]) -> None:
        """One predict_action call for the batch. Batch sizes are rounded up to 1 / 2 / 4 / 8 / 16 … with copies of the
        last request, so the model builds (and its engine LRU holds) few distinct engines."""
        try:
            n = len(batch)
            size = 1
            while size < n:
                size *= 2
            ids = torch.cat([r.input_ids for r in batch] + [batch[-1].input_ids] * (size - n), dim=0)
            pv = torch.cat([r.pixel_values for r in batch] + [batch[-1].pixel_values] * (size - n), dim=0)
            actions = np.asarray(self.vla.predict_action(input_ids=ids, pixel_values=pv, unnorm_key=batch[0].unnorm_key,
                                                         do_sample=False))
            actions = actions.reshape(size, -1)[:n]
            self.batch_sizes.append(n)
            for r, a in zip(batch, actions):
                r.future.set_result(a)
        except Exception as e:   # noqa: BLE001 — delivered to every waiting request
            for r in batch:
                if not r.future.done():
                    r.future.set_exception(e)

    def close(self) -> None:
        self._q.put(None)
        self._worker.join(timeout=10)

    # -- HTTP --
    def build_app(self):
        from fastapi import FastAPI
        from fastapi.responses import JSONResponse
        app = FastAPI()

        def act(payload: Dict[str, Any]):      # sync handler: FastAPI runs it in its thread pool, so requests overlap
            return JSONResponse(self.predict_action(payload))

        app.post("/act")(act)
        self.app = app
        return app

    def run(self, host: str = "0.0.0.0", port: int = 8000) -> None:
        import uvicorn
        uvicorn.run(self.build_app(), host=host, port=port)
