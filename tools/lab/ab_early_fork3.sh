#!/bin/bash
# headline workload: early fork of the side stream (MMLREC_EARLY_WGRAD = position in the backward chain)
one() {
  echo -n "EARLY_WGRAD=$1  "
  MMLREC_EARLY_WGRAD=$1 python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['config'].get('early_fork'), d['loss_check']['ok'])
"
}
MMLREC_EARLY_WGRAD_DEBUG=1 MMLREC_EARLY_WGRAD=auto python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 6 --warmup 2 --no-loss-check 2>&1 | grep "early fork"
for rep in 1 2; do for v in 0 1 2 3 4 5 6 7; do one $v; done; done
