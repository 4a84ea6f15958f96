#!/bin/bash
# kernel trace of the normal (two-stream, HIP-graph) bench: which launches overlap and how long they take side by side
out=${1:-gpurun_out/prof_overlap}
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy "$@" > $out/bench.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gather_vec4" in r["Kernel_Name"]]
# third-from-last step = a graph replay (the last ones are the instrumented eager pass)
a, b = idx[-12], idx[-11]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} -> {e/1e3:9.1f} us (+{(e-s)/1e3:7.1f})  q={r.get('Queue_Id','?'):>2}  {r['Kernel_Name'].replace('void mml::','').replace('mml::','')[:60]}")
print("step span us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
PY
