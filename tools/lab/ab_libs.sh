#!/bin/bash
# same-box A/B of two builds of the library (tools/lab/libs/{old,new}.so): step time and the stand-alone time of the table stream
for rep in 1 2; do for v in old new; do
  echo -n "$v  "
  MMLREC_LIB=$PWD/tools/lab/libs/$v.so python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 40 --warmup 5 --no-loss-check "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['ms_per_step'], d['value'], {n: round(v,4) for n,v in k.items() if 'opt_dense' in n or 'scatter' in n}, d['roofline']['kernel'][:24], d['roofline']['frac'])
"
done; done
