#!/bin/bash
# SQ counters of the cut-once weight-gradient kernel (tools/lab/wg_time.py), one pass per set
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/wg_pmc
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 $R/tools/lab/wg_time.py 3 > $out/p$i.log 2>&1
done
python3 - <<EOF
import csv,glob,collections
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
    for r in csv.DictReader(open(f)):
        if 'wgws' in r['Kernel_Name']:
            a=acc[r['Kernel_Name'][:60]][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,d in acc.items():
        print(k, {c: round(v/n) for c,(v,n) in d.items()})
EOF
