#!/bin/bash
one() {  # label, env value, bench args
  local lab=$1 v=$2; shift 2
  echo -n "$lab EARLY_WGRAD=$v  "
  MMLREC_EARLY_WGRAD=$v python3 bench.py --no-cpu-baseline --no-configs --alt-batch 0 --steps 40 --warmup 5 --no-loss-check "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
lz=d.get('lazy_exact')
print(d['ms_per_step'], d['value'], d['config'].get('early_fork'), [r['ms_per_step_steps_only'] for r in lz['runs']] if lz else '')
"
}
MMLREC_EARLY_WGRAD_DEBUG=1 python3 bench.py --no-cpu-baseline --no-configs --alt-batch 0 --steps 6 --warmup 2 --no-loss-check --workload pepnet_amazon --batch 65536 --no-lazy 2>&1 | grep "early fork"
MMLREC_EARLY_WGRAD_DEBUG=1 python3 bench.py --no-cpu-baseline --no-configs --alt-batch 0 --steps 6 --warmup 2 --no-loss-check --table-update lazy_exact --no-lazy 2>&1 | grep "early fork"
for v in 0 2 4 6 8 10 12; do one pepnet $v --workload pepnet_amazon --batch 65536 --no-lazy; done
for v in 0 2 4 6 8; do one ae30lazy $v --table-update lazy_exact --no-lazy; done
