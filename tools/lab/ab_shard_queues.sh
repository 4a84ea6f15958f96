#!/bin/bash
# forced row-sharded step at world 1: do main and side stream share a hardware queue?
one() {
  echo -n "$1  "
  env $1 MMLREC_BENCH_FORCE_SHARD=1 python3 bench.py --gpus 2 --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d.get('loss_check', {}).get('ok'))
"
}
for rep in 1 2; do
  one "MMLREC_SIDE_PROBE=0"
  one "MMLREC_SIDE_PROBE=1"
  one "GPU_MAX_HW_QUEUES=8"
done
echo -n "unsharded  "; python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
