#!/bin/bash
one() {
  echo -n "TAIL_BLOCKS=$1 U=$2 VARIANT=$3  "
  MMLREC_TAIL_BLOCKS=$1 MMLREC_OPT_U=$2 MMLREC_OPT_VARIANT=$3 python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 60 --warmup 10 --no-loss-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['ms_per_step'], d['value'], {n: round(v,4) for n,v in k.items() if 'opt_dense' in n}, d['roofline']['frac'])
"
}
one 0 4 0 > /dev/null
for rep in 1 2; do
  one 3072 4 0; one 3072 4 1; one 100000 2 0; one 100000 2 1; one 0 4 0; one 0 4 1
done
