#!/usr/bin/env python3
"""A/B builds of the tile GEMM family with a different GROUP_M (row tiles per group of the grouped, XCD-contiguous tile
order): python tools/micro/gemm_group_experiments.py 9 18 32 → tools/micro/build/gm<N>/libbridgelang_hip.so; compare with
tools/ab_lib.sh (results are VALID: the order of tiles changes no arithmetic). The product source keeps one constant."""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT / "bridgelang_amd" / "csrc"
SRCS = ["gemm_bf16.hip", "gemm_fp8.hip", "gemm_skinny.hip", "norm.hip", "attention.hip", "attention_bwd.hip", "glue.hip", "train.hip"]
for n in sys.argv[1:]:
    out = Path(__file__).resolve().parent / "build" / f"gm{n}"
    out.mkdir(parents=True, exist_ok=True)
    text = (CSRC / "gemm_bf16.hip").read_text()
    assert len(re.findall(r"constexpr int GROUP_M = \d+;", text)) == 1
    (out / "gemm_bf16.hip").write_text(re.sub(r"constexpr int GROUP_M = \d+;", f"constexpr int GROUP_M = {int(n)};", text))
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", str(out / "gemm_bf16.hip"), "-o", str(out / "gemm_bf16.o")])
    objs = [str(out / "gemm_bf16.o")] + [str(CSRC / "build" / (s[:-4] + ".o")) for s in SRCS[1:]]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", str(out / "libbridgelang_hip.so")])
    print(n, out / "libbridgelang_hip.so")
