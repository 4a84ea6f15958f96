#!/bin/bash
# End-of-round evidence pass (round 5) on the GPU box; everything lands in gpurun_out/, tools/update_profiles.py and
# tools/make_mfma_csv.py copy the judged pieces into profiles/.  (run `rm -rf gpurun_out/traffic gpurun_out/prof_serial
# gpurun_out/pmc2_gemm` HERE first: gpurun merges new files next to old ones)
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/pmc_traffic.sh gpurun_out/traffic > gpurun_out/pmc.log 2>&1; tail -2 gpurun_out/pmc.log
bash tools/prof_serial.sh gpurun_out/prof_serial > gpurun_out/prof_serial.log 2>&1; tail -4 gpurun_out/prof_serial.log
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-300
bash tools/bench_secondary.sh 2>&1 | tail -22
python3 tools/bench_rows.py --graph > gpurun_out/rows.jsonl 2> gpurun_out/rows.err; cat gpurun_out/rows.jsonl
MMLREC_BENCH_FORCE_SHARD=1 python3 bench.py --gpus 2 --no-cpu-baseline --no-configs --no-lazy > gpurun_out/bench_forced_shard.json 2> gpurun_out/bench_forced_shard.err; tail -1 gpurun_out/bench_forced_shard.json | cut -c1-200
GEMM_MASK=1 GEMM_AMAX_OUT=1 bash tools/pmc2.sh gpurun_out/pmc2_gemm tools/bench_gemm.py > gpurun_out/pmc2_gemm.log 2>&1
GEMM_MASK=1 GEMM_AMAX_OUT=1 python3 tools/bench_gemm.py > gpurun_out/bench_gemm.txt 2>&1; tail -16 gpurun_out/bench_gemm.txt
bash tools/lab/trace_step.sh gpurun_out/tl_one --no-configs > gpurun_out/tl_one.txt 2>&1; tail -1 gpurun_out/tl_one.txt
bash tools/lab/trace_step.sh gpurun_out/tl_lazy --no-configs --table-update lazy_exact > gpurun_out/tl_lazy.txt 2>&1; tail -1 gpurun_out/tl_lazy.txt
