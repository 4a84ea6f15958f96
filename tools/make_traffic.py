#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC, one pass each) of `bench.py` into
profiles/traffic.json: average HBM-side bytes per launch for every kernel label bench.py reports.

Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-B read requests as 64 B for wide coalesced streams, so reads are doubled; WRITE_SIZE is exact for 16-B streaming
stores and float atomics.
usage: make_traffic.py <dir with p_fetch/ p_write/ p_atomic/> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def label(kernel_name, avg_bytes_hint=0):
    k = kernel_name.replace("void ", "").replace("mml::", "")
    m = re.match(r"(gemm_pipe_kernel|gemm_panel_kernel|gemm_ws_kernel|gemm_glds_kernel|gemm_kernel|opt_dense_kernel)<([^>]+)>", k)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2))
    for a, b in (("gather_vec4_kernel", "gather_vec4_kernel"), ("scatter_hash_kernel", "scatter_hash_kernel"),
                 ("scatter_fold_kernel", "scatter_fold_kernel"), ("rows_compact_kernel", "rows_compact_kernel"),
                 ("opt_rows_kernel", "opt_rows_kernel"), ("mark_rows_kernel", "mark_rows_kernel"),
                 ("gate_bwd_fast_kernel", "gate_bwd_kernel"), ("gate_fwd_fast_kernel", "gate_fwd_kernel"), ("gate_fwd_once_kernel", "gate_fwd_kernel"),
                 ("head_fast_kernel", "head_kernel"),
                 ("opt_flat_kernel", "opt_flat_kernel"),
                 ("gemm_nt_kernel", "gemm_nt_kernel"), ("nt_reduce_kernel", "slab_reduce"),
                 ("gemm_os_kernel", "gemm_os_kernel"),
                 ("slab_reduce", "slab_reduce")):
        if k.startswith(a):
            return b
    return None


def collect(dirname, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            lab = label(r["Kernel_Name"])
            if lab:
                per[lab].append(float(r["Counter_Value"]))
    return per


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch = collect(f"{root}/p_fetch", "FETCH_SIZE")
    write = collect(f"{root}/p_write", "WRITE_SIZE")
    atom = collect(f"{root}/p_atomic", "TCC_EA0_ATOMIC")
    res = {}
    for lab in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(lab, []), write.get(lab, [])
        if not fv or not wv:
            continue
        fb, wb = 2 * 1024 * sum(fv) / len(fv), 1024 * sum(wv) / len(wv)
        res[lab] = {"hbm_bytes_per_launch": fb + wb, "read_bytes": fb, "write_bytes": wb, "launches_sampled": len(fv)}
        if lab in atom and atom[lab]:
            res[lab]["atomic_requests_64B"] = sum(atom[lab]) / len(atom[lab])
    res["_method"] = ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_ATOMIC in separate passes over `bench.py --steps 6 "
                      "--warmup 2`; KiB -> bytes, reads x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B), averaged "
                      "over all launches of a kernel symbol (GEMM symbols cover several layer shapes)")
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k != "_method":
            print(f"{k:44s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  (r {v['read_bytes'] / 1e6:8.1f}  w {v['write_bytes'] / 1e6:8.1f})")


if __name__ == "__main__":
    main()
