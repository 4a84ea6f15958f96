#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env MMLREC_HEAD_WGS=$1 python3 bench.py --workload $2 --table-update auto --no-configs --no-cpu-baseline --no-lazy --alt-batch 0 --steps 40 --warmup 5 --no-loss-check 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 HEAD_WGS=$1', d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'head' in k})"; }
for w in pepnet_amazon star_amazon mmoe_ae30; do for n in 2 3 4 2 3; do run $n $w; done; done
