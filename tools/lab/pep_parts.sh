#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --workload pepnet_amazon --steps 40 --warmup 8 --no-cpu-baseline --no-lazy --alt-batch ${2:-0} --no-configs --no-loss-check 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'], round(d['value']/1e6,2), (d.get('alt') or {}).get('ms_per_step'))"; }
for rep in 1 2 3; do
run "MMLREC_GRAD_PARTS=0"
run "MMLREC_GRAD_PARTS=1"
done
run "MMLREC_PEP_FUSE=0" 4096
run "MMLREC_PEP_FUSE=1" 4096
run "MMLREC_PEP_FUSE=0" 8192
run "MMLREC_PEP_FUSE=1" 8192
