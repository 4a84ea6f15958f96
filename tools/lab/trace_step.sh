#!/bin/bash
# kernel trace of one replayed step: per-kernel durations and the idle gaps between them
# usage: tools/lab/trace_step.sh <outdir> <bench args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-lazy --alt-batch 0 --steps 20 --warmup 5 "$@" > $out/bench.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last full step: find the last occurrence of the counter kernel that opens a step
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "counter_kernel" in n]
# steps open with a counter kernel; take the window between the 3rd-last and 2nd-last "opening" counters that are > 10 kernels apart
opens = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 5]
a, b = opens[-3], opens[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f} us  gap {(s - prev_end) / 1e3:5.1f}  {r['Kernel_Name'][:70]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step window {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} kernels")
PY
