#!/bin/bash
# Issue / stall / LDS / L2 counters of the two weight-gradient kernels (gemm_nt_kernel against the tile kernel) over
# tools/lab/nt_time.py.  usage: bash tools/lab/nt_pmc.sh <outdir>
out=${1:-gpurun_out/nt_pmc}
PMC_FILTER=gemm bash tools/pmc2.sh $out tools/lab/nt_time.py 3 > $out.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $out/p5 -- python3 tools/lab/nt_time.py 3 > $out/p5.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $out/p6 -- python3 tools/lab/nt_time.py 3 > $out/p6.log 2>&1
python3 - "$out" <<'PY' >> $out.txt
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p5", "p6"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "gemm" in k:
                print(p, k, {c: f"{v:.4g} ({n[(k, c)]} launches)" for c, v in d.items()})
PY
tail -5 $out/p5.log $out/p6.log >> $out.txt
