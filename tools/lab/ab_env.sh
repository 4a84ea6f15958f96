#!/bin/bash
# Interleaved same-box A/B of bench.py under two environment settings.
# usage: tools/lab/ab_env.sh "<env A>" "<env B>" <reps> <bench args...>     (e.g. "MMLREC_MERGE_WGRAD=0" "MMLREC_MERGE_WGRAD=1" 3 --workload mmoe_ae30)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; reps=$3; shift 3
run() { env $1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check "${@:2}" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], round(d['value'] / 1e6, 2))"; }
for rep in $(seq $reps); do
  run "$A" "$@"
  run "$B" "$@"
done
