#!/bin/bash
# SQ counters of the activation-stationary forward kernel on the L1 launch (tools/bench_gemm.py), one pass per set.
# usage (GPU box): tools/lab/panel_pmc.sh <tag> [env assignments for the bench, e.g. MMLREC_LIB=...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/panel_pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  env "$@" GEMM_MASK=1 GEMM_AMAX_OUT=1 FWD_ONLY=1 CASES='L1 experts+gates' rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 $R/tools/bench_gemm.py > $out/p$i.log 2>&1
done
python3 - <<EOF
import csv,glob,collections
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0.0,0])
    for r in csv.DictReader(open(f)):
        if 'gemm_panel' in r['Kernel_Name'] or 'gemm_pipe_kernel<true, true, 128, 0' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,n) in acc.items(): print("$tag", k, v/n)
EOF
