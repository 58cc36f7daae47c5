"""Multi-GPU training = replicated bf16 weights + sharded optimizer (reference FSDP `shard-grad-op`,
prismatic/training/strategies/fsdp.py:80-86,149-158).

MI355X-first instead of FSDP's module wrapping: every rank keeps the full bf16 weights (15 GB of 288) and runs the same
static forward/backward plans; only the fp32 optimizer state is sharded. The flat gradient buffer is cut into BUCKETS in
backward-completion order (lm_head, decoder layer 31 … 0, projector, plain tensors); as soon as a bucket's last wgrad
has been enqueued its reduce-scatter is issued on a side stream (RCCL over xGMI, overlapping the remaining backward),
each rank runs AdamW on its 1/N slice of every bucket, and the updated bf16 slices are all-gathered in place before the
re-pack. Per step and GPU that is (N-1)/N · (4 B + 2 B) per parameter on the links (fp32 reduce-scatter; 2 B + 2 B with
bf16 reduction, the reference default `reduce_in_full_precision=False`) and no collective on the forward path.

This module is device-agnostic bookkeeping + torch.distributed calls (gloo on CPU in tests/test_sharding_cpu.py,
"nccl" = RCCL on GPUs); all arithmetic on GPUs is the bl_* kernels in step.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class Bucket:
    key: str
    offset: int = 0            # into the flat (bucket-padded) parameter space
    numel: int = 0             # padded to a multiple of ALIGN * world
    decay: bool = True
    members: List[int] = field(default_factory=list)     # unit indices


ALIGN = 8                      # elements: 16 bytes of bf16 per rank slice


def _cast(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dtype-converting copy: the bl_cast_* kernels on the GPU (current stream), torch on CPU tensors (gloo tests)."""
    if src.is_cuda:
        from .. import train_ops
        train_ops.cast(src, dst)
    else:
        dst.copy_(src)


def bucket_key(name: str) -> str:
    """Communication bucket of a GEMM-weight group, from its first HF member name."""
    p = name.split(".")
    if name.startswith("language_model.model.layers."):
        return f"llm.layer{int(p[3]):02d}"
    if name.startswith("language_model."):
        return "llm.lm_head"
    if name.startswith("projector."):
        return "projector"
    if name.startswith("vision_backbone."):
        tower = p[1]
        return f"vision.{tower}.block{int(p[3]):02d}" if p[2] == "blocks" else f"vision.{tower}.stem"
    return "misc"


class ShardLayout:
    """Flat parameter space cut into buckets, each padded so that `world` equal, 16-byte aligned slices tile it."""

    def __init__(self, world: int = 1, rank: int = 0):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.world, self.rank = world, rank
        self.buckets: List[Bucket] = []
        self._open: Optional[Bucket] = None
        self.total = 0

    def begin(self, key: str, decay: bool) -> Bucket:
        self.close()
        self._open = Bucket(key, self.total, 0, decay)
        self.buckets.append(self._open)
        return self._open

    def add(self, numel: int, unit_index: int) -> int:
        """Reserve `numel` elements in the open bucket (each unit starts 16-byte aligned); returns the flat offset."""
        b = self._open
        off = b.offset + b.numel
        b.numel += (numel + 3) // 4 * 4
        b.members.append(unit_index)
        return off

    def close(self) -> None:
        b = self._open
        if b is not None:
            q = ALIGN * self.world
            b.numel = (b.numel + q - 1) // q * q
            self.total = b.offset + b.numel
            self._open = None

    # ---- slices ----
    def shard_numel(self, b: Bucket) -> int:
        return b.numel // self.world

    def shard_range(self, b: Bucket, rank: Optional[int] = None) -> Tuple[int, int]:
        """[lo, hi) of rank's slice of bucket b in the flat space."""
        r = self.rank if rank is None else rank
        n = self.shard_numel(b)
        return b.offset + r * n, b.offset + (r + 1) * n

    def local_offset(self, b: Bucket) -> int:
        """Offset of bucket b's slice inside this rank's shard-local buffers (master / moments)."""
        return b.offset // self.world

    @property
    def local_total(self) -> int:
        return self.total // self.world

    def owner_of(self, flat_index: int) -> int:
        for b in self.buckets:
            if b.offset <= flat_index < b.offset + b.numel:
                return (flat_index - b.offset) // self.shard_numel(b)
        raise IndexError(flat_index)


class ShardComm:
    """The three collectives of a sharded-optimizer step. `group=None` with world 1 makes every call a no-op."""

    def __init__(self, layout: ShardLayout, group=None, reduce_dtype: torch.dtype = torch.float32, force: bool = False):
        """`force=True` issues the collectives even at world size 1 (a one-rank RCCL group: exercises the exact calls,
        views and dtypes of the multi-GPU path on a single-GPU box)."""
        self.layout, self.group = layout, group
        self.reduce_dtype = reduce_dtype
        self.active = layout.world > 1 or force
        if self.active and not dist.is_initialized():
            raise RuntimeError("world > 1 needs an initialised process group")
        self._native_rs = self.active and dist.get_backend(group) == "nccl"

    def reduce_scatter_grads(self, grad: torch.Tensor, b: Bucket, scratch: Optional[torch.Tensor] = None) -> None:
        """SUM-reduce bucket b of the flat gradient over ranks; afterwards this rank's slice holds the reduced values
        (other slices are unspecified). Callers pre-divide the loss gradient by `world`, so SUM is the DDP average.
        reduce_dtype=bf16 sends a bf16 copy (`scratch`, same numel as the bucket) — fsdp.py:139-147's reduce_dtype."""
        if self.active:
            self.reduce_scatter_bucket(grad[b.offset:b.offset + b.numel], b, scratch)

    def reduce_scatter_bucket(self, full: torch.Tensor, b: Bucket, scratch: Optional[torch.Tensor] = None) -> None:
        """The same on a bucket-local tensor `full` ([b.numel] fp32: a range of the flat gradient, or the transient slot a
        parameter-sharded layer's weight gradients were written to): afterwards full[rank·n : (rank+1)·n] holds the sum."""
        if not self.active:
            return
        n = self.layout.shard_numel(b)
        lo = self.layout.rank * n
        if self.reduce_dtype != full.dtype:
            send = scratch[:b.numel]
            _cast(full, send)
            mine = send[lo:lo + n]
            self._rs(mine, send)
            _cast(mine, full[lo:lo + n])
        else:
            self._rs(full[lo:lo + n], full)

    def _rs(self, out: torch.Tensor, full: torch.Tensor) -> None:
        if self._native_rs:
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group)    # in place: out ⊂ full
        else:   # gloo has no reduce_scatter: all-reduce the bucket (CPU tests only)
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """Sum of the per-rank partial squared norms (clip_grad_norm over the sharded gradients)."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather_params(self, full: torch.Tensor, b: Bucket) -> None:
        """In place on `full` = the bf16 staging view of bucket b ([b.numel]): every rank's updated slice → the whole
        bucket on every rank."""
        if not self.active:
            return
        n = self.layout.shard_numel(b)
        mine = full[self.layout.rank * n:(self.layout.rank + 1) * n]
        if self._native_rs:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.layout.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)
            for r, p in enumerate(parts):
                full[r * n:(r + 1) * n].copy_(p)

    def all_gather_into(self, full: torch.Tensor, mine: torch.Tensor) -> None:
        """Every rank's slice `mine` (equal sizes) → `full` (world × mine.numel()), rank order — the per-unit parameter
        gather of parameter-sharded training (FSDP FULL_SHARD's pre-forward / pre-backward all-gather, fsdp.py:84-87)."""
        n = mine.numel()
        assert full.numel() == n * self.layout.world
        if not self.active:
            full.copy_(mine)
        elif self._native_rs:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.layout.world)]
            dist.all_gather(parts, mine.contiguous(), group=self.group)
            for r, p in enumerate(parts):
                full[r * n:(r + 1) * n].copy_(p)

    def gather_full(self, local: torch.Tensor) -> torch.Tensor:
        """Shard-local buffer (layout.local_total) → full flat buffer (checkpointing the fp32 masters)."""
        lay = self.layout
        out = torch.empty(lay.total, dtype=local.dtype, device=local.device)
        for b in lay.buckets:
            n = lay.shard_numel(b)
            mine = local[lay.local_offset(b):lay.local_offset(b) + n]
            if not self.active:
                out[b.offset:b.offset + n].copy_(mine)
                continue
            parts = [torch.empty_like(mine) for _ in range(lay.world)]
            dist.all_gather(parts, mine.contiguous(), group=self.group)
            for r, p in enumerate(parts):
                a, e = lay.shard_range(b, r)
                out[a:e].copy_(p)
        return out


def comm_order(buckets: Sequence[Bucket]) -> List[int]:
    """Bucket indices in backward-completion order: lm_head, decoder layers high → low, projector, vision blocks high →
    low, then the plain-tensor buckets (their last contribution — the embedding gradient — lands at the very end)."""
    def rank_of(i: int):
        k = buckets[i].key
        if k == "llm.lm_head":
            return (0, 0)
        if k.startswith("llm.layer"):
            return (1, -int(k[len("llm.layer"):]))
        if k == "projector":
            return (2, 0)
        if k.startswith("vision."):
            blk = k.rsplit(".", 1)[1]
            return (3, -int(blk[5:]) if blk.startswith("block") else 1)
        return (4, i)
    return sorted(range(len(buckets)), key=rank_of)
