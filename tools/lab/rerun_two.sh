#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {  # name, extra env, args...
  local tag=$1; shift
  local envs=$1; shift
  env $envs timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 50 "$@" 2> gpurun_out/secondary_$tag.err | tail -1 \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); d['tag']='$tag'; print(json.dumps(d))" >> gpurun_out/secondary_two.jsonl \
    || echo "{\"tag\": \"$tag\", \"failed\": true}" >> gpurun_out/secondary_two.jsonl
}
: > gpurun_out/secondary_two.jsonl
run mssm_ae30_b65536 "MMLREC_GEMM_MODE=4" --workload mssm_ae30 --batch 65536
run snr_trans_ae30_b65536 "MMLREC_GEMM_MODE=4" --workload snr_trans_ae30 --batch 65536 --steps 12 --warmup 3
python3 -c "
import json
for l in open('gpurun_out/secondary_two.jsonl'):
    d=json.loads(l); print(d['tag'], d.get('failed') or (round(d['value']/1e6,2), d['ms_per_step']))"
