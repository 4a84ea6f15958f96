#!/bin/bash
# (the MMLREC_DUMMY_STREAMS knob these runs used -- extra streams created in front of the graph capture -- was a lab-only line of
#  trainer.TrainStep.run and has been removed: the runs measured level)
cd $GRAFT_REPO_ROOT
out=gpurun_out/coresident3.txt; : > $out
run() { env $1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check "${@:2}" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], round(d['value'] / 1e6, 2))" | tee -a $out; }
C="MMLREC_NT_PER_CU=1 MMLREC_INNER_FORK=2 MMLREC_FORK_MLP=1"
for rep in 1 2; do
  run "MMLREC_NT_PER_CU=2 MMLREC_INNER_FORK=2" --workload mmoe_ae30
  run "$C" --workload mmoe_ae30
  run "$C GPU_MAX_HW_QUEUES=8" --workload mmoe_ae30
  run "$C GPU_MAX_HW_QUEUES=2" --workload mmoe_ae30
  run "$C MMLREC_DUMMY_STREAMS=1" --workload mmoe_ae30
  run "$C MMLREC_DUMMY_STREAMS=2" --workload mmoe_ae30
  run "$C MMLREC_DUMMY_STREAMS=3" --workload mmoe_ae30
  run "MMLREC_NT_PER_CU=2 MMLREC_INNER_FORK=2 GPU_MAX_HW_QUEUES=8" --workload mmoe_ae30
done
