"""CPU oracle for the OpenVLA hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package; the product
(bridgelang_amd/) never does, and fails loudly when its HIP library is missing instead of falling back to anything here.
"""
from __future__ import annotations

import os
import time


def pick_threads(verbose: bool = False) -> int:
    """Choose the torch intra-op thread count that actually runs fastest on this host. On virtualised hosts (e.g. the
    8-vCPU build container) OpenMP barriers between vCPUs cost ~100 ms, so 1 thread beats 8 by 1000x on small ops;
    on a real multi-core host all cores win. Probes a small mixed workload at {all, half, 1} threads."""
    import torch
    n = os.cpu_count() or 1
    x, a, b = torch.randn(1, 261, 1024), torch.randn(288, 1024), torch.randn(1024, 1024)

    def work():
        y = x.to(torch.bfloat16).to(torch.float32)
        y = (y - y.mean(-1, keepdim=True)) * 2.0
        return (a @ b).sum() + y.sum()

    best, best_t = 1, float("inf")
    for nt in sorted({n, max(1, n // 2), 1}, reverse=True):
        torch.set_num_threads(nt)
        work()
        t0 = time.perf_counter()
        for _ in range(3):
            work()
        dt = time.perf_counter() - t0
        if verbose:
            print(f"[oracle] {nt} threads: {dt / 3 * 1e3:.2f} ms")
        if dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    return best
