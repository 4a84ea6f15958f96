#!/bin/bash
# same-box sweep of the marked dense table update: MMLREC_TAIL_BLOCKS (0 = full grid, plain loop) x MMLREC_OPT_U (chunks in
# flight per thread of the capped form); step time and the stand-alone time of the table stream
one() {
  echo -n "TAIL_BLOCKS=$1 U=$2  "
  MMLREC_TAIL_BLOCKS=$1 MMLREC_OPT_U=$2 python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 40 --warmup 5 --no-loss-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['ms_per_step'], d['value'], {n: round(v,4) for n,v in k.items() if 'opt_dense' in n}, d['roofline']['frac'])
"
}
for rep in 1 2; do
  one 0 4
  for u in 2 4 8; do for c in 2048 3072 4096 100000; do one $c $u; done; done
done
