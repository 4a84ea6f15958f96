#!/bin/bash
# stand-alone kernel durations of one training step: bench.py on ONE stream (no co-running kernels), kernel trace + stats
out=${1:-gpurun_out/prof_serial}
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --serial "$@" > $out/bench.log 2>&1
tail -n 3 $out/bench.log
python3 tools/overlap.py $(ls $out/*/*kernel_trace.csv | head -1)
