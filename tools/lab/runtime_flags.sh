#!/bin/bash
# sweep of HIP runtime flags of the kind HIP_FORCE_DEV_KERNARG was (read at runtime initialisation) over bench.py
cd $GRAFT_REPO_ROOT
out=gpurun_out/runtime_flags.txt; : > $out
run() { env $1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs "${@:2}" 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], round(d['value'] / 1e6, 2), 'loss_ok', d['loss_check']['ok'])
except Exception as e:
    print('$1 FAILED', e)" | tee -a $out; }
for rep in 1 2; do
  run "MMLREC_NOP=1"
  for f in AMD_OPT_FLUSH=0 DEBUG_CLR_SKIP_RELEASE_SCOPE=0 DEBUG_CLR_SKIP_RELEASE_SCOPE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 GPU_STREAMOPS_CP_WAIT=0 GPU_STREAMOPS_CP_WAIT=1 DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_DYNAMIC_QUEUES=1; do
    run "$f"
  done
  run "MMLREC_NOP=1"
done
