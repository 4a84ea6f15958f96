#!/bin/bash
mkdir -p gpurun_out/gsweep
for mode in 0 3; do for bn in 64 128; do
  MMLREC_GEMM_MODE=$mode MMLREC_GEMM_BN=$bn timeout 300 python3 tools/bench_gemm.py > gpurun_out/gsweep/m${mode}_bn${bn}.txt 2>&1
done; done
tail -n 30 gpurun_out/gsweep/*.txt
