#!/bin/bash
# AE-30 with its 63 dense columns (K0 = 303): panel kernel on / off and the narrowed input gradient on / off, interleaved on one box
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --workload mmoe_ae30d --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], round(d['value'] / 1e6, 2))"; }
for rep in 1 2 3; do
  run "MMLREC_GEMM_PANEL=1 MMLREC_GRAD_COLS=1"
  run "MMLREC_GEMM_PANEL=0 MMLREC_GRAD_COLS=1"
  run "MMLREC_GEMM_PANEL=1 MMLREC_GRAD_COLS=0"
  run "MMLREC_GEMM_PANEL=0 MMLREC_GRAD_COLS=0"
done
