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
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "counter_kernel" in n]
# replayed steps: the shortest window between two step-opening counter kernels that holds a whole step
best = None
for k in range(len(starts) - 1):
    for nxt in (1, 2):
        if k + nxt >= len(starts):
            continue
        a, b = starts[k], starts[k + nxt]
        if b - a < 15 or b - a > 160:
            continue
        dur = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
        if best is None or dur < best[0]:
            best = (dur, a, b)
dur, a, b = best
t0 = int(rows[a]["Start_Timestamp"])
prev, busy = t0, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f} us  gap {(s - prev) / 1e3:7.1f}  q={r.get('Queue_Id', '?')}  {r['Kernel_Name'][:64]}")
    busy += e - s
    prev = max(prev, e)
print(f"step window {dur / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, {b - a} kernels")
PY
