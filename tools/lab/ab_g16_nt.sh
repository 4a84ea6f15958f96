#!/bin/bash
# round 6: KuaiRec-32 bf16, weight-gradient tiles 128 x 128 (MMLREC_G16_NT_KW=128) against 128 x 256 (default)
cd $GRAFT_REPO_ROOT
run() { env MMLREC_G16_NT_KW=$1 MMLREC_GEMM_MODE=1 python3 bench.py --workload mmoe_kuairec --no-configs --no-cpu-baseline --no-lazy --alt-batch 0 --table-update auto --steps 30 --warmup 5 --no-loss-check 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms_per_step']; print('KW=$1', d['ms_per_step'], round(d['value']/1e6,2), {a[:40]: b for a, b in k.items() if 'nt_kernel' in a or 'reduce' in a})"; }
for rep in 1 2 3; do
run 128
run 256
done
