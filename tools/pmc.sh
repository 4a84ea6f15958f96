#!/bin/bash
# usage: tools/pmc.sh <outdir> <python-script> [args...]   -- three counter passes, kernel-filtered summary
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p1 -- python3 "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/p2 -- python3 "$@" > $out/p2.log 2>&1
rocprofv3 --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace --output-format csv -d $out/p3 -- python3 "$@" > $out/p3.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, d in agg.items():
            if "gemm" in k or "scatter" in k or "gather" in k or "opt_" in k:
                print(p, k, {c: f"{v:.4g}" for c, v in d.items()})
PY
