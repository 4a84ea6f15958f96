#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and prints, for the last steps, the timeline of kernels (start offset, duration,
queue/stream) to see which launches actually overlap."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last occurrence of the gather kernel = start of the last step
idx = [i for i, r in enumerate(rows) if "gather_vec4" in r["Kernel_Name"]]
start = idx[-2] if len(idx) > 1 else 0
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + 40]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void mml::", "").replace("mml::", "")[:48]
    print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q={r.get('Queue_Id', '?'):>3}  {name}")
