#!/bin/bash
# runs tools/bench_gemm.py against every ablation build in tools/lab
mkdir -p gpurun_out/lab
for f in tools/lab/lib_*.so; do
  n=$(basename $f .so)
  CASES="${CASES:-L1 experts+gates,L2}" GEMM_PLANES="${GEMM_PLANES:-0}" MMLREC_LIB=$PWD/$f timeout 300 python3 tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/lab/$n.txt
  echo "== $n"; cat gpurun_out/lab/$n.txt
done
