#!/bin/bash
# rocprofv3 --kernel-trace --stats of one workload's replayed steps -> <out>/..._kernel_stats.csv (+ the bench line in <out>/bench.log)
# usage: tools/lab/stats_workload.sh <outdir> <bench args...>      (environment, e.g. MMLREC_GEMM_MODE=1, is inherited)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --no-loss-check "$@" > $out/bench.log 2>&1
ls $out/*/*kernel_stats.csv | head -1
