"""Condensed instruction schedule of one kernel from a hipcc -S dump: M = MFMA, L = buffer/global load, D = ds op,
[..] = s_waitcnt, BAR = s_barrier, labels and branches. usage: python tools/isa_schedule.py file.hip mangled-substring"""
import re, subprocess, sys
src, key = sys.argv[1], sys.argv[2]
out = "/tmp/_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", src,
                "-o", out], check=True, capture_output=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
seq = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if t.startswith("v_mfma"): seq.append("M")
    elif t.startswith(("buffer_load", "global_load")): seq.append("Llds" if " lds" in t else "L")
    elif t.startswith(("buffer_store", "global_store")): seq.append("S")
    elif t.startswith("ds_"): seq.append("D")
    elif t.startswith("s_waitcnt"): seq.append("[" + t.replace("s_waitcnt ", "") + "]")
    elif t.startswith(".LBB"): seq.append("\n" + t)
    elif t.startswith(("s_cbranch", "s_branch")): seq.append("<" + t.split()[0][2:] + " " + t.split()[-1] + ">")
    elif t.startswith("s_barrier"): seq.append("BAR")
    elif t.startswith("scratch_"): seq.append("SCRATCH")
print(" ".join(seq))
