#!/usr/bin/env python3
"""Timing-only builds of the whole-sequence attention kernel (attn_seq_kernel): where do its 50-80 us go? RESULTS OF THESE
BUILDS ARE INVALID — `nostage` skips the K/V staging (stale LDS), `nocompute` skips the query-tile loop.

    python tools/micro/attn_experiments.py [nostage nocompute]

writes a patched COPY of bridgelang_amd/csrc/attention.hip under tools/micro/build/attn_<name>/ (the product source holds no
experiment switches), builds the whole library around it and prints the path; time it with
BRIDGELANG_HIP_LIB=<path> python tools/bench_attn.py."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT / "bridgelang_amd" / "csrc"
SRCS = ["gemm_bf16.hip", "gemm_fp8.hip", "gemm_skinny.hip", "norm.hip", "attention.hip", "attention_bwd.hip", "glue.hip", "train.hip"]
PATCHES = {
    # a runtime-false bound keeps the code but never runs it (p.Sq is never negative)
    "nostage": [("for (int base = tid; base < npieces; base += 512 * UNR) {", "for (int base = tid; base < npieces && p.Sq < 0; base += 512 * UNR) {", 2)],
    "nocompute": [("for (int r = 0; r * 8 < nloc; ++r) {", "for (int r = 0; r * 8 < nloc && p.Sq < 0; ++r) {", 1)],
}


def build(name: str) -> Path:
    out = Path(__file__).resolve().parent / "build" / f"attn_{name}"
    out.mkdir(parents=True, exist_ok=True)
    text = (CSRC / "attention.hip").read_text()
    for old, new, count in PATCHES[name]:
        if text.count(old) != count:
            raise SystemExit(f"{name}: expected {count} occurrence(s) of {old!r}, found {text.count(old)} — update PATCHES")
        text = text.replace(old, new)
    (out / "attention.hip").write_text(text)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    objs = []
    for s in SRCS:
        src = out / s if s == "attention.hip" else CSRC / s
        obj = out / (s[:-4] + ".o") if s == "attention.hip" else CSRC / "build" / (s[:-4] + ".o")
        if s == "attention.hip":
            subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", str(src), "-o", str(obj)])
        objs.append(str(obj))
    lib = out / "libbridgelang_hip.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", str(lib)])
    return lib


if __name__ == "__main__":
    for n in sys.argv[1:] or list(PATCHES):
        print(n, build(n))
