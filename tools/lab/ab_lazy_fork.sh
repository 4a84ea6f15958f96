#!/bin/bash
# lazy_exact at B = 65 536: the tail is the weight-gradient GEMMs alone; do they gain from starting beside the backward chain?
one() {
  echo -n "$1  "
  env $1 python3 bench.py --table-update lazy_exact --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 40 --warmup 5 --no-loss-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['config'].get('early_fork'))
"
}
for rep in 1 2; do
  one "MMLREC_MERGE_WGRAD=1 MMLREC_EARLY_WGRAD=0"
  one "MMLREC_MERGE_WGRAD=0 MMLREC_EARLY_WGRAD=0"
  for k in 1 2 3; do one "MMLREC_MERGE_WGRAD=0 MMLREC_EARLY_WGRAD=$k"; done
done
