#!/bin/bash
# The three counter passes of tools/pmc_gemm.py (separate rocprofv3 --pmc runs, no trace domains) for the inference and the
# training row counts, into gpurun_out/<tag>/{fetch,write,mfma} and <tag>_t/…:   tools/pmc_run.sh TAG      (on the GPU box, repo root)
TAG="${1:-pmc}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for M in 4608 9472; do
  SUF=""; [ "$M" = 9472 ] && SUF="_t"
  rocprofv3 --pmc FETCH_SIZE -d "$R/gpurun_out/$TAG$SUF/fetch" -- python3 "$R/tools/pmc_gemm.py" $M > /dev/null 2>> "$R/gpurun_out/$TAG.log" || exit 1
  rocprofv3 --pmc WRITE_SIZE -d "$R/gpurun_out/$TAG$SUF/write" -- python3 "$R/tools/pmc_gemm.py" $M > /dev/null 2>> "$R/gpurun_out/$TAG.log" || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$R/gpurun_out/$TAG$SUF/mfma" -- python3 "$R/tools/pmc_gemm.py" $M > /dev/null 2>> "$R/gpurun_out/$TAG.log" || exit 1
done
find "$R/gpurun_out/$TAG" "$R/gpurun_out/${TAG}_t" -name "*.db" | head -8
