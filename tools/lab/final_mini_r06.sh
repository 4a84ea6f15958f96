#!/bin/bash
# round 6, after the evidence pass: counters of the gate kernels in the headline step, refreshed KuaiRec-32
# bf16 stats (wide weight-gradient tiles), refreshed headline line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gate gpurun_out/prof_k16
PMC_FILTER=gate_ bash tools/pmc2.sh gpurun_out/pmc_gate bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --no-loss-check > gpurun_out/pmc_gate.txt 2>&1; tail -8 gpurun_out/pmc_gate.txt | cut -c1-400
MMLREC_GEMM_MODE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k16 -- python3 bench.py --workload mmoe_kuairec --table-update auto --steps 20 --warmup 3 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --no-loss-check > gpurun_out/prof_k16.log 2>&1; tail -1 gpurun_out/prof_k16.log | cut -c1-160
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-200
