#!/bin/bash
# round 6: PepNet / Amazon-8 at B = 65 536: K7 fusion off / on (products in the GEMM turns) / on with gated heads (default)
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --workload pepnet_amazon --steps 40 --warmup 8 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'], round(d['value']/1e6,2)); [print('   ',k,v) for k,v in d['kernels_ms_per_step'].items()]"; }
for rep in 1 2; do
run "MMLREC_PEP_FUSE=0"
run "MMLREC_PEP_FUSE=1 MMLREC_PEP_GATED_HEAD=0"
run "MMLREC_PEP_FUSE=1"
done
