"""CPU: the C-ABI library loads and exports every symbol the header declares (no compute without a GPU), the weight
layout round-trips HF state dicts, the synthetic generator is deterministic, replicas shard correctly under gloo."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from bridgelang_amd import _lib
    header = (ROOT / "include" / "bridgelang_hip.h").read_text()
    declared = set(re.findall(r"\b(bl_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.bl_abi_version() >= 1 and lib.bl_build_arch() == b"gfx950"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from bridgelang_amd import _lib
    monkeypatch.setenv("BRIDGELANG_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.BridgeLangHipError):
        _lib.load()


def test_ops_reject_cpu_tensors():
    from bridgelang_amd import ops
    a = torch.zeros(16, 64, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        ops.gemm(a, torch.zeros(1, 2, 64, 8, dtype=torch.bfloat16), a)


def test_struct_layout_matches_header():
    """ctypes mirrors of bl_gemm_desc / bl_attn_desc: natural alignment, same size the C compiler computes."""
    import ctypes as C
    from bridgelang_amd import _lib
    src = '#include "%s"\n#include <stdio.h>\nint main(){printf("%%zu %%zu\\n", sizeof(bl_gemm_desc), sizeof(bl_attn_desc));return 0;}' % (ROOT / "include" / "bridgelang_hip.h")
    exe = Path(os.environ.get("TMPDIR", "/tmp")) / "bl_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-o", str(exe)], input=src.encode(), check=True)
    g, a = map(int, subprocess.run([str(exe)], capture_output=True, check=True).stdout.split())
    assert C.sizeof(_lib.GemmDesc) == g and C.sizeof(_lib.AttnDesc) == a


def test_weight_layout_roundtrip_and_packing():
    from bridgelang_amd import weights as W
    dims = W.tiny_dims()
    specs = W.tensor_specs(dims)
    g = torch.Generator().manual_seed(0)
    sd = {s.name: torch.randn(s.shape, generator=g).to(torch.bfloat16) for s in specs}
    w = W.allocate(dims, "cpu").load_state_dict(sd)
    back = w.state_dict()
    assert list(back) == [s.name for s in specs] + [s.name for s in W.passthrough_specs(dims)]   # executed tensors, then pass-through
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    # fragment-major element map: packed[nt][ks][lane][j] = W[16nt + (lane&15)][32ks + 8(lane>>4) + j]
    ow, pk = sd["language_model.model.layers.0.self_attn.o_proj.weight"], w.layers[0].o_w
    for nt, ks, lane, j in [(0, 0, 0, 0), (3, 5, 37, 6), (31, 15, 63, 7)]:
        assert pk[nt, ks, lane, j] == ow[16 * nt + (lane & 15), 32 * ks + 8 * (lane >> 4) + j]
    # gate/up interleave and K padding live in the logical matrix
    from bridgelang_amd.weights import _unpack
    gu = _unpack(w.layers[0].gu_w)
    assert torch.equal(gu[0::2], sd["language_model.model.layers.0.mlp.gate_proj.weight"])
    assert torch.equal(gu[1::2], sd["language_model.model.layers.0.mlp.up_proj.weight"])
    fc2 = _unpack(w.siglip.blocks[0].fc2_w)
    assert fc2.shape == (576, 1088) and (fc2[:, 1072:] == 0).all()
    with pytest.raises(KeyError):
        W.allocate(dims, "cpu").load_state_dict({})


def test_7b_spec_counts():
    from bridgelang_amd import weights as W
    d = W.openvla_7b_dims()
    n = sum(int(np.prod(s.shape)) for s in W.tensor_specs(d))
    # 7.523 B in the checkpoint minus the never-executed last ViT blocks / final norms / attn-pool
    assert 7.4e9 < n < 7.53e9, n
    names = [s.name for s in W.tensor_specs(d)]
    assert "vision_backbone.featurizer.blocks.22.ls2.scale_factor" in names
    assert "vision_backbone.featurizer.blocks.23.norm1.weight" not in names      # tap = depth-2
    assert "vision_backbone.fused_featurizer.blocks.25.mlp.fc2.weight" in names
    assert "language_model.model.layers.31.mlp.down_proj.weight" in names and "projector.fc3.bias" in names


def test_synth_generator_deterministic_and_normalish():
    from oracle import synth as S
    a = S.synth_bf16((1000, 64), 123, 0.0, 0.02)
    b = S.synth_bf16((1000, 64), 123, 0.0, 0.02)
    assert torch.equal(a, b)
    assert abs(a.float().mean().item()) < 5e-4 and abs(a.float().std().item() - 0.02) < 5e-4
    assert not torch.equal(a, S.synth_bf16((1000, 64), 124, 0.0, 0.02))
    # chunked generation == one-shot generation
    x = S.synth_f32(1000, 7, 1.0, 0.02, start=5000)
    y = S.synth_f32(6000, 7, 1.0, 0.02)[5000:]
    assert np.array_equal(x, y)


def test_synth_c_helper_matches_numpy_generator():
    """oracle/synth_c.c (built by oracle/Makefile into oracle/_ref/) is the same generator bit for bit."""
    import subprocess
    from pathlib import Path
    from oracle import synth as S
    root = Path(__file__).resolve().parent.parent
    subprocess.run(["make", "-C", str(root / "oracle")], check=True, capture_output=True)
    S._C = None
    assert S._c_lib() is not None
    for shape, mean, std in (((4096, 257), 0.0, 0.02), ((70001,), 1.0, 0.02), ((300, 1000), 0.0, 1.0), ((256, 512), 0.0, 0.12)):
        a, b = S.synth_bf16(shape, 4242, mean, std, use_c=True), S.synth_bf16(shape, 4242, mean, std, use_c=False)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), shape


def test_decisive_recipe_and_overlay_specs():
    """The second synthetic checkpoint differs from the bench one only in the stated std values, and its lm_head overlay
    (action-token rows) is applied identically by the oracle's synth_state_dict."""
    from bridgelang_amd import weights as W
    from oracle import synth as S
    d = W.tiny_dims()
    a, b = W.tensor_specs(d), W.tensor_specs(d, "decisive")
    assert [x.name for x in a] == [x.name for x in b] and [x.shape for x in a] == [x.shape for x in b]
    changed = {x.name.split(".")[-2] for x, y in zip(a, b) if x != y}
    assert changed == {"proj", "fc2", "o_proj", "down_proj", "embed_tokens"}, changed
    (ov,) = W.synthetic_overlays(d, "decisive")
    assert ov.base == "language_model.lm_head.weight" and ov.row0 == 31744 and ov.shape == (256, d.llm_dim)
    assert W.synthetic_overlays(d, "init") == []
    sd = S.synth_state_dict(b, seed=3, overlays=[ov])
    lm = sd["language_model.lm_head.weight"].float()
    assert abs(lm[31744:32000].std().item() - 0.12) < 5e-3 and abs(lm[:31744].std().item() - 0.02) < 1e-3
    assert abs(lm[32000:].std().item() - 0.02) < 2e-3
    with pytest.raises(ValueError):
        W.tensor_specs(d, "nope")


def test_full_size_fixtures_are_well_formed():
    """tests/golden/cfg1_7b_*.npz (oracle/restate.py at openvla-7b, written by tests/golden/make_cfg_7b.py): shapes,
    id ranges, logits are bf16 values whose argmax is the stored id, gaps are consistent with the stored logits."""
    from pathlib import Path
    gold = Path(__file__).resolve().parent / "golden"
    files = sorted(gold.glob("cfg1_7b_*.npz"))
    assert len(files) >= 2
    for f in files:
        z = np.load(f)
        B, L, seed, wseed = [int(v) for v in z["meta"]]
        assert z["ids"].shape == (B, 7) and z["input_ids"].shape == (B, L)
        assert (z["ids"] >= 0).all() and (z["ids"] < 32064).all()
        if str(z["recipe"]) == "decisive":
            assert ((z["ids"] >= 31744) & (z["ids"] < 32000)).all()
        if "logits_bf16" in z:
            lg = torch.from_numpy(z["logits_bf16"].astype(np.int16)).view(torch.bfloat16).float()
            assert lg.shape == (B, 7, 32064) and np.array_equal(lg.argmax(-1).numpy(), z["ids"])
            t2 = lg.topk(2, dim=-1).values
            assert np.allclose((t2[..., 0] - t2[..., 1]).numpy(), z["top2_gap"])
        else:
            vals = torch.from_numpy(z["topk_vals_bf16"].astype(np.int16)).view(torch.bfloat16).float().numpy()
            for b in range(B):          # the greedy id (first maximal index) is one of the entries holding the top value
                for t in range(7):
                    tied = z["topk_idx"][b, t][vals[b, t] == vals[b, t, 0]]
                    assert z["ids"][b, t] in tied


def _timm_vit_keys(depth, layerscale, cls_reg, attn_pool):
    """state_dict keys of a timm 0.9.10 VisionTransformer(num_classes=0) († from knowledge of timm — SURVEY App. A.1;
    the reference builds the towers at modeling_prismatic.py:78-101 and loads them strictly, prismatic.py:113-116)."""
    keys = (["cls_token", "reg_token"] if cls_reg else []) + ["pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    for i in range(depth):
        for n in ("norm1", "attn.qkv", "attn.proj", "norm2", "mlp.fc1", "mlp.fc2"):
            keys += [f"blocks.{i}.{n}.weight", f"blocks.{i}.{n}.bias"]
        if layerscale:
            keys += [f"blocks.{i}.ls1.gamma", f"blocks.{i}.ls2.gamma"]
    keys += ["norm.weight", "norm.bias"]
    if attn_pool:
        keys += ["attn_pool.latent"] + [f"attn_pool.{n}.{wb}" for n in ("q", "kv", "proj", "norm", "mlp.fc1", "mlp.fc2")
                                        for wb in ("weight", "bias")]
    return set(keys)


def test_exported_checkpoint_has_every_key_the_reference_loads_strictly():
    """A state dict written by this package must carry the never-executed tensors too (last ViT block, final norm,
    SigLIP attention pool): the reference's load is strict. Checked in both layouts, with shapes, at 7B
    dims on a meta-free CPU layout object of the tiny dims and by spec at 7B."""
    from bridgelang_amd import weights as W
    from bridgelang_amd.training import checkpoint as C
    d = W.tiny_dims()
    w = W.allocate(d, "cpu")
    sd = w.state_dict()
    mods = C.to_model_state_dicts(sd, ("vision_backbone", "projector", "llm_backbone"))
    vb = mods["vision_backbone"]
    dino = {k[len("dino_featurizer."):] for k in vb if k.startswith("dino_featurizer.")}
    sig = {k[len("siglip_featurizer."):] for k in vb if k.startswith("siglip_featurizer.")}
    assert dino == _timm_vit_keys(d.dino.depth, True, True, False), dino ^ _timm_vit_keys(d.dino.depth, True, True, False)
    assert sig == _timm_vit_keys(d.siglip.depth, False, False, True), sig ^ _timm_vit_keys(d.siglip.depth, False, False, True)
    assert set(mods["projector"]) == {f"projector.{i}.{wb}" for i in (0, 2, 4) for wb in ("weight", "bias")}
    # round trip: load → state_dict keeps the pass-through tensors' values
    sd2 = {k: torch.randn(v.shape).to(torch.bfloat16) for k, v in sd.items()}
    w.load_state_dict(sd2)
    back = w.state_dict()
    for k in ("vision_backbone.fused_featurizer.attn_pool.kv.weight", f"vision_backbone.featurizer.blocks.{d.dino.depth - 1}.ls2.scale_factor",
              "vision_backbone.featurizer.norm.bias", f"vision_backbone.fused_featurizer.blocks.{d.siglip.depth - 1}.mlp.fc1.weight"):
        assert torch.equal(back[k], sd2[k]), k
    # 7B: the full key set has the public parameter count of openvla-7b (7.54 B)
    d7 = W.openvla_7b_dims()
    n = sum(int(np.prod(sp.shape)) for sp in W.tensor_specs(d7) + W.passthrough_specs(d7))
    assert abs(n - 7.5412e9) < 1e6, n
    shapes = {sp.name: sp.shape for sp in W.passthrough_specs(d7)}
    assert shapes["vision_backbone.fused_featurizer.attn_pool.kv.weight"] == (2304, 1152)
    assert shapes["vision_backbone.fused_featurizer.blocks.26.mlp.fc1.weight"] == (4304, 1152)
    assert shapes["vision_backbone.featurizer.blocks.23.attn.qkv.weight"] == (3072, 1024)


def test_shard_properties():
    from bridgelang_amd.replicas import shard
    for n in (0, 1, 16, 17, 100):
        for world in (1, 2, 3, 8):
            parts = [list(shard(n, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["BL_ROOT"])
from bridgelang_amd import replicas
assert replicas.init("gloo")
rank, _, world = replicas.env_rank()
mine = list(replicas.shard(17, rank, world))
replicas.fence()
t = replicas.max_over_ranks(1.0 + rank)
ids = replicas.gather_ids(torch.tensor([rank * 100 + len(mine)]))
print(f"RESULT {rank} {world} {t} {mine[0]} {mine[-1]} {[int(x) for x in ids]}", flush=True)
torch.distributed.destroy_process_group()
'''


def test_replicas_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, BL_ROOT=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = sorted(l for o in outs for l in o.splitlines() if l.startswith("RESULT"))
    assert lines[0] == "RESULT 0 2 2.0 0 8 [9, 108]", lines
    assert lines[1] == "RESULT 1 2 2.0 9 16 [9, 108]", lines


def test_bench_launcher_starts_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus 2` (no torchrun): the parent starts 2 ranks that rendezvous on 127.0.0.1, time exactly K
    steps between fences, take the max over ranks, and rank 0 prints ONE JSON line with n_gpus = 2 (dry run: gloo, no
    GPU). The same command under torchrun must accept the ranks it is given, and a --gpus / WORLD_SIZE mismatch is an error."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["ms_per_step"] >= 19.0                      # rank 1 sleeps 20 ms per step: the MAX over ranks is reported
    assert abs(d["value"] - 2 * 16 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29577", str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_registries_and_checkpoint_naming():
    """Reference ids resolve (models/materialize.py:29-73), unknown ids raise its ValueError; native ↔ HF key maps are
    inverse bijections over every tensor of the model (convert_openvla_weights_to_hf.py:73-115)."""
    from bridgelang_amd.models.materialize import LLM_BACKBONES, VISION_BACKBONES, get_llm_backbone_and_tokenizer, get_vision_backbone_and_transform
    from bridgelang_amd.training.checkpoint import from_model_state_dicts, hf_to_prismatic, prismatic_to_hf, to_model_state_dicts
    from bridgelang_amd import weights as W
    assert "dinosiglip-vit-so-224px" in VISION_BACKBONES and "llama2-7b-pure" in LLM_BACKBONES
    vb, tf = get_vision_backbone_and_transform("dinosiglip-vit-so-224px", "resize-naive")
    assert (vb.embed_dim, vb.num_patches, vb.default_image_resolution) == (2176, 256, (3, 224, 224)) and callable(tf)
    lb, tok = get_llm_backbone_and_tokenizer("llama2-7b-pure", llm_max_length=2048, tokenizer="tok")
    assert tok == "tok" and lb.embed_dim == 4096 and lb.pad_token_id == 32000 and lb.prompt_builder_fn.__name__ == "PurePromptBuilder"
    assert lb.last_layer_finetune_modules[1] == "language_model.model.layers.31"
    with pytest.raises(ValueError):
        get_vision_backbone_and_transform("clip-vit-l", "resize-naive")
    with pytest.raises(ValueError):
        get_llm_backbone_and_tokenizer("phi-2-3b")
    with pytest.raises(RuntimeError):
        vb.forward(torch.zeros(1, 6, 224, 224))                  # not bound to a VLM
    names = [s.name for s in W.tensor_specs(W.tiny_dims())]
    seen = set()
    for n in names:
        m, k = hf_to_prismatic(n)
        assert prismatic_to_hf(m, k) == n and (m, k) not in seen
        seen.add((m, k))
    assert hf_to_prismatic("projector.fc2.bias") == ("projector", "projector.2.bias")
    assert hf_to_prismatic("vision_backbone.featurizer.blocks.0.ls1.scale_factor") == ("vision_backbone", "dino_featurizer.blocks.0.ls1.gamma")
    assert hf_to_prismatic("language_model.lm_head.weight") == ("llm_backbone", "llm.lm_head.weight")
    sd = {n: torch.zeros(1) for n in names}
    assert set(from_model_state_dicts(to_model_state_dicts(sd, ["vision_backbone", "projector", "llm_backbone"]))) == set(names)


@pytest.mark.parametrize("H,W", [(256, 256), (480, 640), (100, 300), (224, 500), (300, 224), (128, 128), (224, 224), (1080, 1920)])
def test_resample_coefficient_tables_match_pillow(H, W):
    """ops.resample_coeffs (the host half of bl_resample_pass_u8) + the numpy restatement of the two 8-bit passes give
    Pillow's bicubic resize bit for bit (down- and up-sampling, one-axis-only cases, the identity)."""
    from bridgelang_amd import ops
    from oracle import resample as RS
    rng = np.random.default_rng(H * 10007 + W)
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    frames[1, : H // 2] = 255                      # a hard edge: bicubic over/undershoot hits the clip8 path
    frames[1, H // 2:] = 0
    tw = None if W == 224 else tuple(t.numpy() for t in ops.resample_coeffs(W, 224)[:2])
    th = None if H == 224 else tuple(t.numpy() for t in ops.resample_coeffs(H, 224)[:2])
    got = RS.two_pass(frames, tw, th)
    want = RS.pil_resize(frames, 224, 224)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("H,W,oh,ow", [(256, 256, 224, 224), (480, 640, 224, 224), (128, 128, 224, 224), (224, 300, 224, 224), (256, 256, 128, 160)])
def test_lanczos_coefficient_tables_match_pillow(H, W, oh, ow):
    """The Lanczos-3 tables (ops.resample_coeffs(filter="lanczos"), the host half of the device twin of LIBERO's
    resize_image, libero_utils.py:33-47) through the numpy restatement of the two 8-bit passes ≡ Pillow's
    `Image.resize(LANCZOS)` bit for bit."""
    from PIL import Image
    from bridgelang_amd import ops
    from oracle import resample as RS
    rng = np.random.default_rng(H * 31 + W)
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    frames[1, : H // 2] = 255
    frames[1, H // 2:] = 0
    tw = None if W == ow else tuple(t.numpy() for t in ops.resample_coeffs(W, ow, "lanczos")[:2])
    th = None if H == oh else tuple(t.numpy() for t in ops.resample_coeffs(H, oh, "lanczos")[:2])
    got = RS.two_pass(frames, tw, th)
    want = np.stack([np.asarray(Image.fromarray(f).resize((ow, oh), Image.LANCZOS), dtype=np.uint8) for f in frames])
    assert np.array_equal(got, want)


def test_crop_sampling_constants_reproduce_the_host_grid():
    """eval_preprocess.sampling_constants (what bl_crop_resize_bilinear_u8 is launched with) regenerates, in fp32, the
    sampling grid of the host restatement of tf.image.crop_and_resize for the centre-crop box."""
    from bridgelang_amd.vla import eval_preprocess as EP
    for (H, W), out_hw, cs in (((256, 256), (224, 224), 0.9), ((480, 640), (224, 224), 0.9), ((224, 224), (224, 224), 1.0),
                               ((100, 77), (13, 1), 0.5)):
        box = EP.center_crop_box(cs)
        yb, ys, xb, xs = EP.sampling_constants(box, (H, W), out_hw)
        oh, ow = out_hw
        f32 = np.float32
        assert all(f32(v) == v for v in box), "the box is computed in fp32 (tf.sqrt / clip / offsets on float32 tensors)"
        side = np.clip(np.sqrt(f32(cs)), f32(0), f32(1))
        assert box == (float((f32(1) - side) / f32(2)),) * 2 + (float((f32(1) - side) / f32(2) + side),) * 2
        y1, x1, y2, x2 = (f32(v) for v in box)
        # TF's CropAndResize kernel: scale = (y2 - y1) * (H - 1) / (out - 1); in_y = y1 * (H - 1) + i * scale, all fp32
        want_y = y1 * f32(H - 1) + np.arange(oh, dtype=np.float32) * ((y2 - y1) * f32(H - 1) / f32(max(oh - 1, 1)))
        want_x = x1 * f32(W - 1) + np.arange(ow, dtype=np.float32) * ((x2 - x1) * f32(W - 1) / f32(max(ow - 1, 1)))
        if ow == 1:
            want_x = np.array([f32(0.5) * (x1 + x2) * f32(W - 1)], dtype=np.float32)
        got_y = np.float32(yb) + np.arange(oh, dtype=np.float32) * np.float32(ys)
        got_x = np.float32(xb) + np.arange(ow, dtype=np.float32) * np.float32(xs)
        assert np.array_equal(got_y, want_y) and np.array_equal(got_x, want_x)
