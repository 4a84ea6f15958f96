#!/bin/bash
# usage: tools/pmc2.sh <outdir> <python-script> [args...]   -- stall / issue counters of the GEMM kernels, three passes
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $out/p1 -- python3 "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --kernel-trace --output-format csv -d $out/p2 -- python3 "$@" > $out/p2.log 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p3 -- python3 "$@" > $out/p3.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/p4 -- python3 "$@" > $out/p4.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in agg.items():
            if os.environ.get("PMC_FILTER", "gemm") in k:
                print(p, k, {c: f"{v:.4g}" for c, v in d.items()})
PY
