#!/bin/bash
# End-of-round evidence pass (round 6) on the GPU box; everything lands in gpurun_out/, tools/update_profiles.py and
# tools/make_mfma_csv.py copy the judged pieces into profiles/.  (run `rm -rf gpurun_out/traffic gpurun_out/prof_serial
# gpurun_out/pmc2_gemm gpurun_out/prof_pep gpurun_out/prof_k16 gpurun_out/pmc_scatter` HERE first: gpurun merges new files
# next to old ones)
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/pmc_traffic.sh gpurun_out/traffic > gpurun_out/pmc.log 2>&1; tail -2 gpurun_out/pmc.log
bash tools/prof_serial.sh gpurun_out/prof_serial > gpurun_out/prof_serial.log 2>&1; tail -4 gpurun_out/prof_serial.log
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-300
bash tools/bench_secondary.sh 2>&1 | tail -22
python3 tools/bench_rows.py --graph > gpurun_out/rows.jsonl 2> gpurun_out/rows.err; cat gpurun_out/rows.jsonl
MMLREC_BENCH_FORCE_SHARD=1 python3 bench.py --gpus 2 --no-cpu-baseline --no-configs --no-lazy > gpurun_out/bench_forced_shard.json 2> gpurun_out/bench_forced_shard.err; tail -1 gpurun_out/bench_forced_shard.json | cut -c1-200; grep preflight gpurun_out/bench_forced_shard.err
GEMM_MASK=1 GEMM_AMAX_OUT=1 bash tools/pmc2.sh gpurun_out/pmc2_gemm tools/bench_gemm.py > gpurun_out/pmc2_gemm.log 2>&1
GEMM_MASK=1 GEMM_AMAX_OUT=1 python3 tools/bench_gemm.py > gpurun_out/bench_gemm.txt 2>&1; tail -16 gpurun_out/bench_gemm.txt
bash tools/lab/trace_step.sh gpurun_out/tl_one --no-configs > gpurun_out/tl_one.txt 2>&1; tail -1 gpurun_out/tl_one.txt
bash tools/lab/trace_step.sh gpurun_out/tl_lazy --no-configs --table-update lazy_exact > gpurun_out/tl_lazy.txt 2>&1; tail -1 gpurun_out/tl_lazy.txt
bash tools/lab/trace_step.sh gpurun_out/tl_4k_lazy --no-configs --no-loss-check --batch 4096 --table-update lazy_exact > gpurun_out/tl_4k_lazy.txt 2>&1; tail -1 gpurun_out/tl_4k_lazy.txt
bash tools/lab/trace_step.sh gpurun_out/tl_pep --no-configs --no-loss-check --workload pepnet_amazon --table-update auto > gpurun_out/tl_pep.txt 2>&1; tail -1 gpurun_out/tl_pep.txt
# per-kernel stats of the two secondary configurations the round worked on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pep gpurun_out/prof_k16
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pep -- python3 bench.py --workload pepnet_amazon --table-update auto --steps 20 --warmup 3 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --no-loss-check > gpurun_out/prof_pep.log 2>&1; tail -1 gpurun_out/prof_pep.log | cut -c1-160
MMLREC_GEMM_MODE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k16 -- python3 bench.py --workload mmoe_kuairec --table-update auto --steps 20 --warmup 3 --no-cpu-baseline --no-configs --alt-batch 0 --no-lazy --no-loss-check > gpurun_out/prof_k16.log 2>&1; tail -1 gpurun_out/prof_k16.log | cut -c1-160
# the scatter's counters: stand-alone launches of tools/bench_rows.py
PMC_FILTER=scatter bash tools/pmc2.sh gpurun_out/pmc_scatter tools/bench_rows.py --batches 65536 --dists zipf --reps 3 > gpurun_out/pmc_scatter.txt 2>&1; tail -6 gpurun_out/pmc_scatter.txt
