#!/bin/bash
# (run `rm -rf gpurun_out/traffic gpurun_out/prof_serial` HERE first: gpurun merges new files next to old ones)
# End-of-round evidence pass on the GPU box: smoke, PMC traffic, serial kernel stats, the default bench line, secondary
# workloads.  Everything lands in gpurun_out/; tools/update_profiles.py copies the judged pieces into profiles/.
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/pmc_traffic.sh gpurun_out/traffic > gpurun_out/pmc.log 2>&1; tail -2 gpurun_out/pmc.log
bash tools/prof_serial.sh gpurun_out/prof_serial > gpurun_out/prof_serial.log 2>&1; tail -4 gpurun_out/prof_serial.log
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-300
bash tools/bench_secondary.sh 2>&1 | tail -14
python3 tools/bench_rows.py --graph > gpurun_out/rows.jsonl 2> gpurun_out/rows.err; cat gpurun_out/rows.jsonl
MMLREC_BENCH_FORCE_SHARD=1 python3 bench.py --gpus 2 --no-cpu-baseline --no-configs --no-lazy > gpurun_out/bench_forced_shard.json 2> gpurun_out/bench_forced_shard.err; tail -1 gpurun_out/bench_forced_shard.json | cut -c1-200
