#!/bin/bash
# weight gradients at ONE workgroup per CU beside the table scatter / optimizer: does the other branch become resident?
# usage: tools/lab/coresident.sh  -> gpurun_out/coresident.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/coresident.txt; : > $out
run() { env $1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check "${@:2}" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], round(d['value'] / 1e6, 2))" | tee -a $out; }
for rep in 1 2; do
  run "MMLREC_NT_PER_CU=2 MMLREC_INNER_FORK=2" --workload mmoe_ae30
  run "MMLREC_NT_PER_CU=1 MMLREC_INNER_FORK=2" --workload mmoe_ae30
  run "MMLREC_NT_PER_CU=1 MMLREC_INNER_FORK=3" --workload mmoe_ae30
  run "MMLREC_NT_PER_CU=1 MMLREC_INNER_FORK=2 MMLREC_FORK_MLP=1" --workload mmoe_ae30
  run "MMLREC_NT_PER_CU=2 MMLREC_INNER_FORK=2 MMLREC_FORK_MLP=1" --workload mmoe_ae30
  run "MMLREC_NT_PER_CU=1 MMLREC_INNER_FORK=0" --workload mmoe_ae30
done
MMLREC_NT_PER_CU=1 bash tools/lab/trace_step.sh gpurun_out/trace_nt1 --no-configs --no-loss-check 2>&1 | tail -26 | tee -a $out
