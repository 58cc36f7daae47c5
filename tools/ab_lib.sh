#!/bin/bash
# A/B two builds of the library on the Llama prefill GEMM shapes, alternating within ONE gpurun call (box-to-box noise is
# larger than most kernel-schedule effects):  tools/ab_lib.sh /path/libA.so /path/libB.so [rounds]
A="$1"; B="$2"; N="${3:-2}"
SH="4608,12288,4096 4608,4096,4096,res 4608,4096,11008,res 4608,22016,4096"
for i in $(seq $N); do
  for L in "$A" "$B"; do
    echo "== $L"; BRIDGELANG_HIP_LIB="$L" python tools/bench_gemm_shapes.py $SH 2>/dev/null
  done
done
