#!/usr/bin/env python3
"""Copies the judged profile summaries from gpurun_out/ (scratch) into profiles/ (tracked).
usage: update_profiles.py <traffic dir> <serial prof dir> [round tag]"""
import collections
import csv
import glob
import os
import shutil
import sys

traffic, prof = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(os.path.join(out, f"{tag}_pmc"), exist_ok=True)
shutil.copy(os.path.join(traffic, "traffic.json"), os.path.join(out, "traffic.json"))
for p, c in (("p_fetch", "FETCH_SIZE"), ("p_write", "WRITE_SIZE"), ("p_atomic", "TCC_EA0_ATOMIC")):
    fs = sorted(glob.glob(f"{traffic}/{p}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]
    if not fs:
        continue
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            e = d[r["Kernel_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    with open(os.path.join(out, f"{tag}_pmc", f"{p}_summary.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "AvgPerDispatch"])
        for k, (n, s) in sorted(d.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, n, s, s / n])
ks = sorted(glob.glob(f"{prof}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
if ks:  # (gpurun merges new files next to old ones: take the newest)
    shutil.copy(ks[-1], os.path.join(out, f"{tag}_kernel_stats.csv"))
log = os.path.join(prof, "bench.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        open(os.path.join(out, f"{tag}_kernel_stats_bench_line.json"), "w").write(lines[-1])
gout = os.path.join(root, "gpurun_out")
for src, dst in (("secondary.jsonl", f"{tag}_secondary.jsonl"), ("bench_default.json", f"{tag}_bench_line.json")):
    p = os.path.join(gout, src)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(out, dst), "w").write("".join(lines if src.endswith("jsonl") else lines[-1:]))
print("profiles updated:", sorted(os.listdir(out)))
