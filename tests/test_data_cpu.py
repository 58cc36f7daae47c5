"""CPU: image processor, processor, sample construction (RLDSBatchTransform / DummyDataset) → collator: the host-side
chain that produces the tensors the device path consumes."""
from types import SimpleNamespace

import numpy as np
import torch
from PIL import Image


class WordTokenizer:
    """Deterministic stand-in for the Llama tokenizer (none exists offline): BOS + one id per whitespace-separated piece;
    integer pieces (what the stub's decode() emits for action ids) map back to themselves."""
    vocab_size, bos_token_id, eos_token_id, pad_token_id = 32000, 1, 2, 32000

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)

    def batch_decode(self, rows):
        return [self.decode(r) for r in rows]

    def _ids(self, text):
        out = [1]
        for w in text.replace("</s>", " </s>").split():
            out.append(2 if w == "</s>" else int(w) if w.isdigit() else 3 + (sum(map(ord, w)) % 20000))
        return out

    def __call__(self, text, add_special_tokens=True, return_tensors=None, **_):
        if isinstance(text, str) and return_tensors is None:
            return SimpleNamespace(input_ids=self._ids(text))
        rows = [self._ids(t) for t in ([text] if isinstance(text, str) else text)]
        ids = torch.tensor(rows)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}


def test_image_processor_layout_and_values():
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor, PrismaticProcessor
    g = np.random.RandomState(0)
    img = Image.fromarray(g.randint(0, 256, (224, 224, 3), dtype=np.uint8))     # already 224: resize is the identity
    ip = PrismaticImageProcessor()
    pv = ip.apply_transform(img)
    assert pv.shape == (6, 224, 224) and pv.dtype == torch.float32
    x = torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float() / 255
    m, s = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    assert torch.allclose(pv[:3], (x - m) / s, atol=1e-6) and torch.allclose(pv[3:], (x - 0.5) / 0.5, atol=1e-6)
    big = Image.fromarray(g.randint(0, 256, (561, 772, 3), dtype=np.uint8))     # the fork's test.jpg geometry
    out = ip(big, return_tensors="pt")["pixel_values"]
    assert out.shape == (1, 6, 224, 224) and torch.isfinite(out).all()
    proc = PrismaticProcessor(ip, WordTokenizer())
    enc = proc("In: What action should the robot take to grasp the snack bag?\nOut:", big)
    assert enc.pixel_values.shape == (1, 6, 224, 224) and enc["input_ids"].shape == enc.attention_mask.shape
    cast = enc.to("cpu", dtype=torch.bfloat16)
    assert cast.pixel_values.dtype == torch.bfloat16 and cast.input_ids.dtype == torch.int64


def test_sample_construction_and_collation():
    from bridgelang_amd.extern.hf.processing_prismatic import PrismaticImageProcessor
    from bridgelang_amd.util.data_utils import PaddedCollatorForActionPrediction
    from bridgelang_amd.vla.action_tokenizer import ActionTokenizer
    from bridgelang_amd.vla.datasets import DummyDataset, RLDSBatchTransform
    tok = WordTokenizer()
    at = ActionTokenizer(tok)
    ip = PrismaticImageProcessor()
    ds = DummyDataset(at, tok, ip.apply_transform, seed=0)
    a, b = ds[0], ds[0]
    assert torch.equal(a["input_ids"], b["input_ids"]) and torch.equal(a["pixel_values"], b["pixel_values"])
    ids, labels = a["input_ids"], a["labels"]
    assert ids[0] == 1 and ids[-1] == 2                                            # BOS … </s>
    assert (labels[:-8] == -100).all() and torch.equal(labels[-8:], ids[-8:])      # 7 action tokens + stop token
    assert ((ids[-8:-1] > at.action_token_begin_idx) & (ids[-8:-1] < 32000)).all() # action ids live in the last 256
    # the encoded ids round-trip to the bin centres of the original action
    act = np.asarray(np.random.RandomState(0).rand(224 * 224 * 3 + 7)[-7:], dtype=np.float32)
    dec = at.decode_token_ids_to_actions(ids[-8:-1].numpy())
    assert np.all(np.abs(dec - np.clip(act, -1, 1)) <= 2.0 / 255 + 1e-6)
    rl = RLDSBatchTransform(at, tok, ip.apply_transform, predict_stop_token=False)
    s = rl({"dataset_name": "bridge_orig", "action": act[None], "task": {"language_instruction": b"Put The Carrot On The Plate"},
            "observation": {"image_primary": np.zeros((1, 224, 224, 3), dtype=np.uint8)}})
    assert s["dataset_name"] == "bridge_orig" and s["labels"][-1] == -100 and (s["labels"][-8:-1] != -100).all()
    batch = PaddedCollatorForActionPrediction(tok.vocab_size and 2048, tok.pad_token_id)([ds[0], ds[1], s])
    assert batch["pixel_values"].shape == (3, 6, 224, 224) and batch["input_ids"].shape == batch["labels"].shape
    assert torch.equal(batch["attention_mask"], batch["input_ids"].ne(32000))


def test_eval_time_preprocessing_properties():
    """openvla_utils.py:81-155 / libero_utils.py:33-47 restated without TensorFlow (absent: parity unpinned): defining
    properties of the centre crop + resize and of the JPEG / Lanczos resize."""
    from bridgelang_amd.vla.eval_preprocess import (center_crop_and_resize, crop_and_resize_bilinear, jpeg_round_trip,
                                                    resize_image)
    g = np.random.RandomState(0)
    img = g.randint(0, 256, (224, 224, 3), dtype=np.uint8)
    # crop_scale = 1 at the same size: sample points land on the pixel centres → the image itself (× 255.5 truncation)
    same = center_crop_and_resize(img, crop_scale=1.0, out_hw=(224, 224))
    assert same.shape == (224, 224, 3) and same.dtype == np.uint8 and np.array_equal(same, img)
    # a constant image stays constant; the crop is centred: a centred bright square covers more of the output
    const = np.full((100, 120, 3), 77, np.uint8)
    assert (center_crop_and_resize(const, 0.9) == 77).all()
    sq = np.zeros((200, 200, 3), np.uint8); sq[50:150, 50:150] = 255
    out = center_crop_and_resize(sq, 0.9, (200, 200))
    assert out[100, 100, 0] == 255 and out[2, 2, 0] == 0
    frac_in, frac_out = (sq > 127).mean(), (out > 127).mean()
    assert abs(frac_out / frac_in - 1 / 0.9) < 0.03                       # area ratio = 1 / crop_scale
    # corners of the box are sampled exactly (crop_and_resize uses out-1 intervals)
    ramp = np.tile(np.arange(101, dtype=np.float32)[None, :, None], (5, 1, 1))
    r = crop_and_resize_bilinear(ramp, (0.0, 0.1, 1.0, 0.9), (5, 9))
    assert np.allclose(r[0, :, 0], np.linspace(10, 90, 9), atol=1e-4)
    assert (crop_and_resize_bilinear(ramp, (0.0, -0.5, 1.0, 0.5), (2, 3))[0, 0] == 0).all()    # outside → 0
    # JPEG round trip is lossy but close on a smooth image; Lanczos resize keeps range / shape / dtype
    yy, xx = np.mgrid[0:256, 0:256]
    smooth = np.stack([yy, xx, (yy + xx) // 2], -1).astype(np.uint8)
    jt = jpeg_round_trip(smooth)
    assert jt.shape == smooth.shape and np.abs(jt.astype(int) - smooth.astype(int)).mean() < 2.0
    rs = resize_image(smooth, (224, 224))
    assert rs.shape == (224, 224, 3) and rs.dtype == np.uint8 and abs(int(rs[112, 112, 0]) - 128) <= 3
