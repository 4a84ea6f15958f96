#!/bin/bash
# Same-box sweep of the CU partition knobs (trainer.py: MMLREC_CU_TAIL, MMLREC_CU_EARLY) on bench.py.
# usage: tools/lab/cu_split.sh "<tail values>" "<early values>"  -> gpurun_out/cu_split.log
mkdir -p gpurun_out
out=gpurun_out/cu_split.log
: > $out
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); a=d.get("alt") or {}
        print(d["ms_per_step"], a.get("ms_per_step"), {k:v for k,v in list(d["kernels_ms_per_step"].items())[:4]})'
for t in ${1:-0 64 96 128 160}; do
  echo "== CU_TAIL=$t" >> $out
  MMLREC_CU_TAIL=$t python bench.py --no-cpu-baseline --no-lazy --steps 30 2>>gpurun_out/cu_split.err | python -c "$pick" >> $out
done
for e in ${2:-}; do
  echo "== CU_EARLY=$e (B=4096 split)" >> $out
  MMLREC_CU_EARLY=$e python bench.py --no-cpu-baseline --no-lazy --steps 30 --batch 4096 --alt-batch 0 2>>gpurun_out/cu_split.err | python -c "$pick" >> $out
done
cat $out
