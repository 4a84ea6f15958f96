#!/bin/bash
# HBM traffic counters of the bench, one rocprofv3 --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass)
out=${1:-gpurun_out/traffic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
B="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --serial"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p_fetch -- python3 $B > $out/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/p_write -- python3 $B > $out/p_write.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC --kernel-trace --output-format csv -d $out/p_atomic -- python3 $B > $out/p_atomic.log 2>&1
python3 tools/make_traffic.py $out $out/traffic.json
