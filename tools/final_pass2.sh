#!/bin/bash
# Second half of the end-of-round evidence pass: MFMA / issue counters of the GEMM kernels, the GEMM table, step timelines.
rm -rf gpurun_out/pmc2_gemm
GEMM_MASK=1 GEMM_AMAX_OUT=1 bash tools/pmc2.sh gpurun_out/pmc2_gemm tools/bench_gemm.py > gpurun_out/pmc2_gemm.log 2>&1
GEMM_MASK=1 GEMM_AMAX_OUT=1 python3 tools/bench_gemm.py > gpurun_out/bench_gemm.txt 2>&1; tail -16 gpurun_out/bench_gemm.txt
bash tools/lab/trace_step.sh gpurun_out/tl_serial --no-configs --serial > gpurun_out/tl_serial.txt 2>&1; tail -1 gpurun_out/tl_serial.txt
bash tools/lab/trace_step.sh gpurun_out/tl_2s --no-configs > gpurun_out/tl_2s.txt 2>&1; tail -1 gpurun_out/tl_2s.txt
bash tools/lab/trace_step.sh gpurun_out/tl_lazy --no-configs --table-update lazy_exact > gpurun_out/tl_lazy.txt 2>&1; tail -1 gpurun_out/tl_lazy.txt
