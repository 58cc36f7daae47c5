"""CPU (gloo, world size 2): the sharded-optimizer bookkeeping of the training step — bucket layout, reduce-scatter of
gradients, per-slice AdamW, all-gather of the updated weights, master gather — reproduces a single-process AdamW step
on the rank-averaged gradients (what FSDP `shard-grad-op` computes, fsdp.py:80-86). The arithmetic stand-in for
bl_adamw_f32 here is the torch formula (the kernel itself is checked on the GPU in test_train_ops_gpu.py)."""
import os
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent


def test_layout_slices_tile_every_bucket():
    from bridgelang_amd.training.sharding import ShardLayout, bucket_key, comm_order
    for world in (1, 2, 3, 8):
        lay = ShardLayout(world, 0)
        sizes = {"llm.layer00": [100, 37], "llm.layer01": [64], "llm.lm_head": [1001], "plain.nodecay": [5, 7, 9]}
        ui = 0
        offs = []
        for key, units in sizes.items():
            lay.begin(key, not key.endswith("nodecay"))
            for n in units:
                offs.append((lay.add(n, ui), n))
                ui += 1
        lay.close()
        assert lay.total % (8 * world) == 0 and lay.local_total * world == lay.total
        for (o, n), (o2, _) in zip(offs, offs[1:]):
            assert o + n <= o2 and o % 4 == 0                       # units do not overlap, 16-byte aligned fp32
        for b in lay.buckets:
            edges = [lay.shard_range(b, r) for r in range(world)]
            assert edges[0][0] == b.offset and edges[-1][1] == b.offset + b.numel
            assert all(a[1] == c[0] for a, c in zip(edges, edges[1:]))
            assert all((hi - lo) % 8 == 0 for lo, hi in edges)
        order = [lay.buckets[i].key for i in comm_order(lay.buckets)]
        assert order == ["llm.lm_head", "llm.layer01", "llm.layer00", "plain.nodecay"], order
    assert bucket_key("language_model.model.layers.7.mlp.down_proj.weight") == "llm.layer07"
    assert bucket_key("language_model.lm_head.weight") == "llm.lm_head"
    assert bucket_key("projector.fc2.weight") == "projector"
    assert bucket_key("vision_backbone.featurizer.blocks.3.attn.qkv.weight") == "vision.featurizer.block03"
    assert bucket_key("vision_backbone.fused_featurizer.patch_embed.proj.weight") == "vision.fused_featurizer.stem"


_WORKER = r'''
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["BL_ROOT"])
from bridgelang_amd import replicas
from bridgelang_amd.training.sharding import ShardLayout, ShardComm, comm_order
assert replicas.init("gloo")
rank, _, world = replicas.env_rank()
reduce_dtype = torch.bfloat16 if os.environ["BL_REDUCE"] == "bf16" else torch.float32

def build(world, rank):
    lay = ShardLayout(world, rank)
    ui = 0
    for key, units, decay in (("llm.layer00", [1000, 333], True), ("llm.lm_head", [4097], True), ("plain.nodecay", [17, 5], False)):
        lay.begin(key, decay)
        for n in units:
            lay.add(n, ui); ui += 1
    lay.close()
    return lay

def adamw(p, m, v, g, step, lr, wd, coef):
    g = g * coef
    p.mul_(1 - lr * wd)
    m.mul_(0.9).add_(g, alpha=0.1)
    v.mul_(0.999).addcmul_(g, g, value=0.001)
    p.addcdiv_(m / (1 - 0.9 ** step), (v / (1 - 0.999 ** step)).sqrt() + 1e-8, value=-lr)

lay = build(world, rank)
ref = build(1, 0)           # NOTE: padding differs with world; compare per unit through offsets below
comm = ShardComm(lay, reduce_dtype=reduce_dtype)
g0 = torch.Generator().manual_seed(0)
init = torch.randn(lay.total, generator=g0)                       # same on every rank
master_full = init.clone()
master = torch.cat([master_full[slice(*lay.shard_range(b))] for b in lay.buckets])
m, v = torch.zeros_like(master), torch.zeros_like(master)
stage = master_full.to(torch.bfloat16)
ref_p, ref_m, ref_v = init.clone(), torch.zeros(lay.total), torch.zeros(lay.total)
for step in (1, 2):
    grads = [torch.randn(lay.total, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
    grad = grads[rank].clone() / world                            # loss gradient pre-divided by world (step.py)
    scratch = torch.empty(lay.total, dtype=torch.bfloat16)
    for i in comm_order(lay.buckets):
        b = lay.buckets[i]
        comm.reduce_scatter_grads(grad, b, scratch[b.offset:b.offset + b.numel])
    partial = torch.zeros(len(lay.buckets))
    for i, b in enumerate(lay.buckets):
        lo, hi = lay.shard_range(b)
        partial[i] = (grad[lo:hi] ** 2).sum()
    comm.all_reduce_sum(partial)
    norm = partial.sum().sqrt()
    coef = min(1.0, 1.0 / (norm.item() + 1e-6))
    for b in lay.buckets:
        lo, hi = lay.shard_range(b)
        sl = slice(lay.local_offset(b), lay.local_offset(b) + hi - lo)
        adamw(master[sl], m[sl], v[sl], grad[lo:hi], step, 1e-2, 0.1 if b.decay else 0.0, coef)
        stage[lo:hi] = master[sl].to(torch.bfloat16)
        comm.all_gather_params(stage[b.offset:b.offset + b.numel], b)
    # single-process reference on the averaged gradients
    if reduce_dtype == torch.bfloat16:
        avg = sum((g / world).to(torch.bfloat16).float() for g in grads)      # bf16 wire format, fp32 accumulate ≈
    else:
        avg = sum(g / world for g in grads)
    rn = avg.norm()
    rc = min(1.0, 1.0 / (rn.item() + 1e-6))
    for b in lay.buckets:
        sl = slice(b.offset, b.offset + b.numel)
        adamw(ref_p[sl], ref_m[sl], ref_v[sl], avg[sl], step, 1e-2, 0.1 if b.decay else 0.0, rc)
    tol = 2e-2 if reduce_dtype == torch.bfloat16 else 1e-5
    assert abs(norm.item() - rn.item()) <= tol * rn.item(), (norm.item(), rn.item())
    full = comm.gather_full(master)
    err = (full - ref_p).abs().max().item()
    assert err <= (2e-2 if reduce_dtype == torch.bfloat16 else 1e-5), err
    assert torch.equal(stage.float(), full.to(torch.bfloat16).float())
print(f"RESULT {rank} ok {lay.total} {lay.local_total}", flush=True)
dist.destroy_process_group()
'''


def _run(tmp_path, reduce, port):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, BL_ROOT=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", BL_REDUCE=reduce)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = sorted(l for o in outs for l in o.splitlines() if l.startswith("RESULT"))
    assert len(lines) == 2 and all(" ok " in l for l in lines), outs


def test_sharded_adamw_gloo_world2_fp32(tmp_path):
    _run(tmp_path, "fp32", 29541)


def test_sharded_adamw_gloo_world2_bf16_reduce(tmp_path):
    _run(tmp_path, "bf16", 29542)


def test_full_shard_covers_every_fsdp_unit_of_the_reference():
    """fsdp-full-shard's bookkeeping on the CPU (no kernels): with `shard_params` the ParamStore of a full fine-tune shards the
    parameters AND gradients of every FSDP unit the reference wraps — each ViT block and patch embedding, the projector, every
    decoder layer, and the root's token embeddings / lm_head (prismatic.py:285-306, dinosiglip_vit.py:136-140, fsdp.py:160-168);
    the sharded buckets are one contiguous range of the flat space, the rank keeps 1/world of them, and only the small plain
    tensors (norm scales, biases, position / class tokens) stay replicated. Vision frozen (vla-train): the vision units drop out,
    the rest is unchanged."""
    from bridgelang_amd.training.step import ParamStore, param_sharded_key, pool_of
    from bridgelang_amd.weights import allocate, tiny_dims
    dims = tiny_dims()
    w = allocate(dims, "cpu")
    for world in (1, 2):
        for stage in ("vla-full-train", "vla-train"):
            st = ParamStore(w, stage, world, world - 1, shard_params=True, defer_grads=True)
            keys = [b.key for b in st.layout.buckets]
            sharded = [k for k in keys if k in st.sharded_keys]
            assert all(param_sharded_key(k) for k in sharded) and not any(param_sharded_key(k) for k in keys if k not in st.sharded_keys)
            assert {"projector", "llm.lm_head", "llm.embed"} <= set(sharded)
            assert sum(k.startswith("llm.layer") for k in sharded) == dims.llm_layers
            n_vis = sum(k.startswith("vision.") for k in sharded)
            assert n_vis == (dims.dino.n_run + dims.siglip.n_run + 2 if stage == "vla-full-train" else 0)
            assert {pool_of(k) for k in sharded} == ({"layers", "head", "vision"} if n_vis else {"layers", "head"})
            lo, hi = st._cut                                      # contiguous: [first sharded bucket, last sharded bucket)
            assert sum(b.numel for b in st.layout.buckets if b.key in st.sharded_keys) == hi - lo
            rest = [k for k in keys if k not in st.sharded_keys]
            assert set(rest) <= {"plain.decay", "plain.nodecay"}
            replicated = sum(b.numel for b in st.layout.buckets if b.key not in st.sharded_keys)
            assert replicated <= 0.02 * st.layout.total
            assert st.own.numel() * world <= (hi - lo) + 8 and st.stage_bf16.numel() <= replicated + 8
            # two gradient-slot parities per pool, alternating along each pool's unit order
            for pool in {pool_of(k) for k in sharded}:
                par = [st.grad_slot_of[k][1] for k in keys if k in st.sharded_keys and pool_of(k) == pool and pool != "layers"]
                assert par == [i % 2 for i in range(len(par))]
