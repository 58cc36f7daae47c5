import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    from oracle import pick_threads
    pick_threads()   # 1 thread on virtualised hosts (OpenMP barriers cost ~100 ms there), all cores on real ones


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def bf(x):
    """fp32 CPU tensor (bf16-representable values) → bf16."""
    import torch
    return x.to(torch.bfloat16)


def rand_bf16(shape, seed, scale=1.0):
    """Seeded CPU tensor of bf16-representable fp32 values."""
    import torch
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).float()
