"""Condense the rocprofv3 --pmc passes of tools/pmc_gemm.py (one directory per pass, each with p_counter_collection.csv)
into profiles/pmc_rNN/: the bl_* kernel rows as small CSVs + gemm_traffic.json + mfma_util.json.
    python tools/pmc_summarize.py gpurun_out/pmc_r02 profiles/pmc_r02 [M [out-name.json]]   (M: rows of the driver run, default 4608)"""
import csv, json, sys, collections
from pathlib import Path
src, dst = Path(sys.argv[1]), Path(sys.argv[2])
dst.mkdir(parents=True, exist_ok=True)
csv.field_size_limit(1 << 30)
rows = collections.defaultdict(list)          # counter → [(dispatch, kernel, grid, value, duration)]
for f in sorted(src.glob("*/p_counter_collection.csv")):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "bl_" not in k:
                continue
            k = k.replace("void ", "").split("(")[0]
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), k, int(r["Grid_Size"]), float(r["Counter_Value"]),
                                            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for f in sorted(src.glob("*/*/*_results.db")):            # rocprofv3's default output since ROCm 7: one rocpd database per pass
    import sqlite3
    cur = sqlite3.connect(f).cursor()
    for d, k, g, c, v, t0, t1 in cur.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, start, end "
                                             "from counters_collection order by dispatch_id"):
        if "bl_" not in k:
            continue
        rows[c].append((int(d), k.replace("void ", "").split("(")[0], int(g), float(v), int(t1) - int(t0)))
for c, rs in rows.items():
    with open(dst / f"{c.lower()}_gemm_rows.csv", "w") as fh:
        fh.write("Dispatch_Id,Kernel,Grid_Size,Counter_Name,Counter_Value,DurationNs\n")
        for d, k, g, v, t in rs:
            if "gemm" in k:
                fh.write(f"{d},{k},{g},{c},{v:.1f},{t}\n")
# the driver launches qkv, o, gate_up, down three times each, in that order: group GEMM dispatches per call
M = int(sys.argv[3]) if len(sys.argv) > 3 else 16 * 288
OUT_NAME = sys.argv[4] if len(sys.argv) > 4 else "gemm_traffic.json"
shapes = [("qkv N=12288 K=4096", 12288, 4096, 1.0), ("o N=4096 K=4096", 4096, 4096, 1.0),
          ("gate_up N=22016 K=4096", 22016, 4096, 0.5), ("down N=4096 K=11008", 4096, 11008, 1.0)]


def per_call(counter):
    """kernels of one bl_gemm_bf16 call (main launch + tail launch) summed; median over the 3 calls per shape"""
    rs = [r for r in rows[counter] if "gemm" in r[1] and "pack" not in r[1]]
    calls, cur = [], []
    for r in rs:                                  # a tail kernel belongs to the preceding main kernel
        if "tail" in r[1] and cur:
            cur.append(r)
        else:
            if cur:
                calls.append(cur)
            cur = [r]
    if cur:
        calls.append(cur)
    return calls


traffic = {"how": ("separate passes on one MI355X: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 tools/pmc_gemm.py "
                   "(bl_* rows beside this file). Units: KB → bytes. gfx950 correction: FETCH_SIZE x2 for wide coalesced reads "
                   "(guides/MI355X_MICROARCH.md, HBM); the counters include Infinity-Cache hits. WRITE_SIZE is uncalibrated for "
                   "8-byte-per-lane stores."),
           "workload": f"Llama-2-7B GEMM shapes at M={M} rows, one bl_gemm_bf16 call each, median of 3 calls",
           "per_call": {}}
fc, wc = per_call("FETCH_SIZE"), per_call("WRITE_SIZE")
tot_h = tot_a = 0.0
for i, (name, N, K, outfrac) in enumerate(shapes):
    f = sorted(sum(x[3] for x in c) for c in fc[3 * i:3 * i + 3])[1]
    w = sorted(sum(x[3] for x in c) for c in wc[3 * i:3 * i + 3])[1]
    kern = "+".join(sorted({x[1].split("::")[-1] for x in fc[3 * i]}))
    algo = 2.0 * (M * K + N * K + M * N * outfrac) + (2.0 * M * N if name.startswith(("o ", "down")) else 0.0)
    hbm = f * 1024 * 2 + w * 1024
    traffic["per_call"][name] = {"kernels": kern, "fetch_kb_raw": f, "write_kb": w, "hbm_bytes_corrected": hbm,
                                 "algorithmic_bytes": algo, "ratio": round(hbm / algo, 2)}
    tot_h += hbm; tot_a += algo
traffic["avg_hbm_bytes_per_call_llama_layer"] = tot_h / 4
traffic["avg_algorithmic_bytes_per_call_llama_layer"] = tot_a / 4
traffic["ratio"] = tot_h / tot_a
(dst / OUT_NAME).write_text(json.dumps(traffic, indent=1))
if rows.get("SQ_VALU_MFMA_BUSY_CYCLES"):
    by = lambda c: {r[0]: r for r in rows[c]}
    mf, gr = by("SQ_VALU_MFMA_BUSY_CYCLES"), by("GRBM_GUI_ACTIVE")
    util = collections.defaultdict(list)
    clock = []
    for d, r in mf.items():
        if "gemm" not in r[1] or d not in gr:
            continue
        cyc = gr[d][3] / 8.0                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        util[f"{r[1].split('::')[-1]} grid={r[2]}"].append(r[3] / (cyc * 1024))
        clock.append(cyc / gr[d][4])
    out = {"how": ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/pmc_gemm.py; MfmaUtil = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); shader clock = GRBM_GUI_ACTIVE / 8 / duration"),
           "shader_clock_GHz_during_gemm": round(sum(clock) / len(clock), 3),
           "mfma_util": {k: round(sorted(v)[len(v) // 2], 4) for k, v in util.items()}}
    (dst / "mfma_util.json").write_text(json.dumps(out, indent=1))
print(json.dumps(traffic["per_call"], indent=1))
print((dst / "mfma_util.json").read_text() if (dst / "mfma_util.json").exists() else "")
