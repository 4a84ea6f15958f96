crashes=0
for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests -m gpu -x -q > /tmp/pt.log 2>&1; rc=$?; if [ $rc -ne 0 ]; then crashes=$((crashes+1)); cp /tmp/pt.log gpurun_out/crash_head.log; fi; done
echo "crashes=$crashes/10 last: $(grep 'passed\|failed' /tmp/pt.log | tail -1)"
