"""Print VGPR / spill / occupancy / LDS of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py bridgelang_amd/csrc/gemm_skinny.hip [name-filter]"""
import re, subprocess, sys
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
for b in re.split(r"remark: Function Name: ", r.stderr)[1:]:
    name = b.split("\n")[0].split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt not in dn:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    occ, lds = g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dn[:100]:100s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} occ {occ} LDS {lds}")
