#!/bin/bash
# B = 4096 split dense update: early pass on a CU-masked stream, U=4 variant of the streaming loop (MMLREC_OPT_VARIANT=4)
mkdir -p gpurun_out
out=gpurun_out/cu_early.log
: > $out
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l)
        print(d["ms_per_step"], {k:v for k,v in list(d["kernels_ms_per_step"].items())[:2]})'
for v in 0 4; do for e in ${1:-0 128 160 192 224}; do
  echo "== VARIANT=$v CU_EARLY=$e" >> $out
  MMLREC_OPT_VARIANT=$v MMLREC_CU_EARLY=$e python bench.py --no-cpu-baseline --no-lazy --steps 50 --batch 4096 --alt-batch 0 2>>gpurun_out/cu_early.err | python -c "$pick" >> $out
done; done
cat $out
