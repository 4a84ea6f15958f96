#!/bin/bash
# same-box A/B of the early side-stream fork (MMLREC_EARLY_WGRAD) on the workloads whose tail has no long table stream
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
for v in 0 auto 0 auto; do one pepnet $v --workload pepnet_amazon --batch 65536 --no-lazy; done
for v in 0 auto; do one star $v --workload star_amazon --batch 65536 --no-lazy; done
for v in 0 auto; do one ae30 $v; done
for v in 0 auto; do one kuairec $v --workload mmoe_kuairec --batch 65536 --no-lazy; done
