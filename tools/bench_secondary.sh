#!/bin/bash
# SURVEY 8(d) "Secondary": the same pure-step measurement for BASELINE configs[1], [2], [4] on one MI355X.
# One JSON line per run -> gpurun_out/secondary.jsonl (tools/update_profiles.py copies it to profiles/).
mkdir -p gpurun_out
out=gpurun_out/secondary.jsonl
: > $out
run() {  # name, extra env, args...
  local tag=$1; shift
  local envs=$1; shift
  env $envs timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-lazy --alt-batch 0 --steps 50 "$@" 2> gpurun_out/secondary_$tag.err | tail -1 \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); d['tag']='$tag'; print(json.dumps(d))" >> $out \
    || echo "{\"tag\": \"$tag\", \"failed\": true}" >> $out
}
for B in 4096 65536; do
  run kuairec_f32_b$B  "MMLREC_GEMM_MODE=4" --workload mmoe_kuairec --batch $B
  run kuairec_bf16_b$B "MMLREC_GEMM_MODE=1" --workload mmoe_kuairec --batch $B
  run ple_ijcai_b$B    "MMLREC_GEMM_MODE=4" --workload ple_ijcai --batch $B
  run star_amazon_b$B  "MMLREC_GEMM_MODE=4" --workload star_amazon --batch $B
  run pepnet_amazon_b$B "MMLREC_GEMM_MODE=4" --workload pepnet_amazon --batch $B
  run ae30d_b$B        "MMLREC_GEMM_MODE=4" --workload mmoe_ae30d --batch $B
done
for w in mlp_ae30 esmm_ae30 cross_stitch_ae30 hmoe_ae30 aitm_ae30 mssm_ae30; do
  run ${w}_b65536 "MMLREC_GEMM_MODE=4" --workload $w --batch 65536
done
# SNR-trans on two alternating batches overfits (loss 0.004 by step 31) and then diverges (all-NaN input gradients by step
# 41: every run() above takes 5 + 50 + 10 steps): 12 timed steps keep the measurement inside the training regime
run snr_trans_ae30_b65536 "MMLREC_GEMM_MODE=4" --workload snr_trans_ae30 --batch 65536 --steps 12 --warmup 3
python3 - <<'PY'
import json
for l in open('gpurun_out/secondary.jsonl'):
    d = json.loads(l)
    if d.get('failed'):
        print(d['tag'], 'FAILED'); continue
    r = d['roofline']
    print(f"{d['tag']:22s} {d['value']/1e6:8.2f} M samples/s  {d['ms_per_step']:.3f} ms  dominant {r['kernel'][:48]} frac {r['frac']}" + ("" if d.get("gradients_finite", True) else "  DIVERGED (non-finite gradients)"))
PY
