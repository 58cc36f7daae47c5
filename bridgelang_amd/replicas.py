"""Multi-GPU inference = independent replicas: sequences are independent and the 15 GB model fits one
288 GB MI355X many times over, so the batch is sharded across ranks with NO data-path collective. The only
communication is the benchmark/serving fence: a barrier and a max-over-ranks of the elapsed time. One process per GPU
(torchrun); backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> bool:
    """Join the process group if launched under torchrun (WORLD_SIZE > 1). Returns True when distributed."""
    _, _, world = env_rank()
    if world <= 1:
        return False
    if not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return True


def shard(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of range(n_items) for `rank` (first n_items % world ranks get one extra)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


def fence(device: Optional[torch.device] = None) -> None:
    """barrier bracketed by device synchronisation (the bench.py timing contract)."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ids(local: torch.Tensor) -> List[torch.Tensor]:
    """Collect per-rank result tensors on every rank (serving convenience; not on the timed path)."""
    if not dist.is_initialized():
        return [local]
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local)
    return out
