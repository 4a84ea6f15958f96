#!/bin/bash
# round 6: soak of the fork INSIDE the step's graph (MMLREC_INNER_FORK=2: a multi-branch HIP graph, the
# kind whose launch segfaulted sporadically in hip::Graph::UpdateStreams).  N fresh processes of the suite's
# graph tests (every model of the zoo builds its own streams: the original repro's shape) and N of bench.py, fork forced on.
# usage: fork_soak.sh [N=10] [fork=2] [bench=1]     (logs of failing runs: gpurun_out/soak_fail_*.log)
cd $GRAFT_REPO_ROOT
N=${1:-10}
export MMLREC_INNER_FORK=${2:-2}
BENCH=${3:-1}
ok_t=0; ok_b=0
for i in $(seq $N); do
  python3 -X faulthandler -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -q -x -p no:cacheprovider -k "fused_train_steps or bench_configuration or bench_secondary or bf16_bench or sharded_path or prefetch" > /tmp/soak_t_$i.log 2>&1
  rc=$?; grep -E "passed|failed|error" /tmp/soak_t_$i.log | tail -1 | sed "s/^/tests run $i rc=$rc: /"
  if [ $rc -eq 0 ]; then ok_t=$((ok_t+1)); else cp /tmp/soak_t_$i.log gpurun_out/soak_fail_fork${MMLREC_INNER_FORK}_$i.log; echo "tests run $i rc=$rc (log kept)"; fi
  if [ "$BENCH" = "1" ]; then
  python3 bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-lazy > /tmp/soak_b_$i.json 2> /tmp/soak_b_$i.err
  rc=$?; python3 -c "
import json,sys
try:
    d=json.loads(open('/tmp/soak_b_$i.json').read().strip().splitlines()[-1]); print('bench run $i rc=$rc:', d['ms_per_step'], 'ms', round(d['value']/1e6,2), 'M samples/s')
except Exception as e:
    print('bench run $i rc=$rc: no line', e)"
  [ $rc -eq 0 ] && ok_b=$((ok_b+1))
  fi
done
echo "SOAK: tests $ok_t / $N clean, bench $ok_b / $N clean (MMLREC_INNER_FORK=$MMLREC_INNER_FORK)"
