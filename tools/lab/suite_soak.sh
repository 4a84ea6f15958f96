#!/bin/bash
# N fresh-process runs of the whole GPU suite on 16 cores (what the driver runs once): a flaky test is a red round
cd $GRAFT_REPO_ROOT
N=${1:-5}
ok=0
for i in $(seq $N); do
  taskset -c 0-15 python3 -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/suite_$i.log 2>&1
  rc=$?; grep -E "passed|failed|error" /tmp/suite_$i.log | tail -1 | sed "s/^/suite run $i rc=$rc: /"
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else cp /tmp/suite_$i.log gpurun_out/suite_fail_$i.log; grep -E "^FAILED|Fatal|Aborted|Segmentation" /tmp/suite_$i.log | head -5; fi
done
echo "SUITE SOAK: $ok / $N clean"
