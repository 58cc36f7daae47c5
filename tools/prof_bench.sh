#!/bin/bash
# rocprofv3 kernel trace of one bench.py invocation, summarised into gpurun_out/<tag>_kernel_stats.csv + <tag>_line.json:
#   tools/prof_bench.sh TAG [bench.py flags…]      (run on the GPU box from the repo root)
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" > "$OUT/${TAG}_line.json" 2> "$OUT/${TAG}_err.txt"
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python3 "$GRAFT_REPO_ROOT/tools/rocprof_summary.py" "$DB" "$OUT/${TAG}_kernel_stats.csv" "${@:0:0}" > "$OUT/${TAG}_summary.txt" 2>&1
tail -30 "$OUT/${TAG}_summary.txt"
