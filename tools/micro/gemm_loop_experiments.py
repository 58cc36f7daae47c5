#!/usr/bin/env python3
"""Timing-only builds of the 256x256 staggered GEMM loop (DESIGN §4 "where the loop's time goes"). RESULTS OF THESE
BUILDS ARE INVALID — they drop synchronisation or staging on purpose to see what a K-tile costs without it.

    python tools/micro/gemm_loop_experiments.py nobar|nowait|noissue|noread [...]

writes a patched COPY of bridgelang_amd/csrc/gemm_bf16.hip under tools/micro/build/<name>/ (the product source holds no
experiment switches), builds the whole library around it as tools/micro/build/<name>/libbridgelang_hip.so and prints the
path; run a bench against it with BRIDGELANG_HIP_LIB=<path> (tools/bench_gemm.py, tools/ab_lib.sh)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT / "bridgelang_amd" / "csrc"
SRCS = ["gemm_bf16.hip", "gemm_fp8.hip", "gemm_skinny.hip", "norm.hip", "attention.hip", "attention_bwd.hip", "glue.hip",
        "train.hip"]

BAR = ("#define BAR()                                   \\\n  do {                                          \\\n"
       "    __builtin_amdgcn_s_barrier();               \\\n    __builtin_amdgcn_sched_barrier(0);          \\\n  } while (0)\n")
PATCHES = {
    # every in-loop s_barrier of gemm256s_kernel dropped: how much of the loop is barrier cost
    "nobar": [(BAR, "#define BAR() __builtin_amdgcn_sched_barrier(0)\n")],
    # no vmcnt wait: loop time without waiting for the LDS-DMA
    "nowait": [('#define WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")\n',
                '#define WAIT_VM8() asm volatile("" ::: "memory")\n'),
               ('#define WAIT_VM10_LGKM() asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory")\n',
                '#define WAIT_VM10_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")\n')],
    # segment cost without the LDS-DMA issue (stale LDS contents) / without the fragment reads
    "noissue": [("#define ISSUE_READ(I, R) do { R; __builtin_amdgcn_sched_barrier(0); I; } while (0)\n",
                 "#define ISSUE_READ(I, R) do { R; } while (0)\n")],
    "noread": [("#define ISSUE_READ(I, R) do { R; __builtin_amdgcn_sched_barrier(0); I; } while (0)\n",
                "#define ISSUE_READ(I, R) do { I; } while (0)\n")],
}


def build(name: str) -> Path:
    out = Path(__file__).resolve().parent / "build" / name
    out.mkdir(parents=True, exist_ok=True)
    text = (CSRC / "gemm_bf16.hip").read_text()
    # only the gemm256s section (the LAST definition of each macro before its #undef block) is patched
    for old, new in PATCHES[name]:
        at = text.rfind(old)
        if at < 0:
            raise SystemExit(f"{name}: macro text not found in gemm_bf16.hip — update PATCHES")
        text = text[:at] + new + text[at + len(old):]
    (out / "gemm_bf16.hip").write_text(text)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    objs = []
    for s in SRCS:
        src = out / s if s == "gemm_bf16.hip" else CSRC / s
        obj = out / (s[:-4] + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", str(src), "-o", str(obj)])
        objs.append(str(obj))
    lib = out / "libbridgelang_hip.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", str(lib)])
    return lib


if __name__ == "__main__":
    names = sys.argv[1:] or list(PATCHES)
    for n in names:
        if n not in PATCHES:
            raise SystemExit(f"unknown experiment {n}: choose from {', '.join(PATCHES)}")
        print(n, build(n))
