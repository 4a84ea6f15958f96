#!/usr/bin/env python3
"""profiles/<tag>_mfma.csv from the four rocprofv3 --pmc passes of tools/pmc2.sh (gpurun_out/<dir>/p1..p4): per GEMM
kernel the summed counters, dispatches, and the derived figures (MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
(4 SIMDs x SQ_BUSY_CU_CYCLES); VALU instructions per MFMA).  usage: make_mfma_csv.py <pmc2 outdir> <tag>"""
import collections
import csv
import glob
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in ("p1", "p2", "p3", "p4"):
    # (gpurun merges a run's files next to those of earlier runs: the newest file of each pass only)
    for f in sorted(glob.glob(f"{src}/{p}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]:
                continue
            e = agg[r["Kernel_Name"]][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
out = os.path.join(root, "profiles", f"{tag}_mfma.csv")
with open(out, "w", newline="") as o:
    o.write('"# rocprofv3 --pmc (four separate passes, tools/pmc2.sh) over tools/bench_gemm.py (grouped GEMMs of MMoE / AE-30 at '
            'M = 65536 and a 4096^3 square; magnitudes measured outside the timed launches); sums over all SQ instances; '
            'derived rows: MFMA_PIPE_BUSY = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), VALU_PER_MFMA = '
            '(SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA"\n')
    w = csv.writer(o)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "AvgPerDispatch"])
    for k, d in sorted(agg.items()):
        for c, (n, s) in sorted(d.items()):
            w.writerow([k, c, n, s, s / n])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CU_CYCLES" in d:
            w.writerow([k, "MFMA_PIPE_BUSY", "", round(d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4 * d["SQ_BUSY_CU_CYCLES"][1]), 4), ""])
        if "SQ_INSTS_VALU" in d and d.get("SQ_INSTS_MFMA", [0, 0])[1] > 0:
            w.writerow([k, "VALU_PER_MFMA", "", round((d["SQ_INSTS_VALU"][1] - d["SQ_INSTS_MFMA"][1]) / d["SQ_INSTS_MFMA"][1], 2), ""])
print("wrote", out)
