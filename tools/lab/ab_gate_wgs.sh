#!/bin/bash
# KuaiRec-32 bf16: workgroups per CU of the gate row kernels (H = 256: one sample per wave and trip)
cd $GRAFT_REPO_ROOT
run() { env $1 MMLREC_GEMM_MODE=1 python3 bench.py --workload mmoe_kuairec --no-configs --no-cpu-baseline --no-lazy --alt-batch 0 --table-update auto --steps 30 --warmup 5 --no-loss-check 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms_per_step']; print('$1', d['ms_per_step'], round(d['value']/1e6,2), {a[:24]: b for a, b in k.items() if 'gate' in a})"; }
run "MMLREC_GATE_FWD_WGS=4 MMLREC_GATE_BWD_WGS=4"
run "MMLREC_GATE_FWD_WGS=6 MMLREC_GATE_BWD_WGS=4"
run "MMLREC_GATE_FWD_WGS=8 MMLREC_GATE_BWD_WGS=4"
run "MMLREC_GATE_FWD_WGS=4 MMLREC_GATE_BWD_WGS=6"
run "MMLREC_GATE_FWD_WGS=4 MMLREC_GATE_BWD_WGS=8"
run "MMLREC_GATE_FWD_WGS=3 MMLREC_GATE_BWD_WGS=3"
run "MMLREC_GATE_FWD_WGS=2 MMLREC_GATE_BWD_WGS=2"
