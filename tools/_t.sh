for i in 1 2; do python3 bench.py --no-cpu-baseline --no-lazy --alt-batch 0 --steps 10 --serial 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms_per_step']['gate_fwd_kernel'], d['kernels_ms_per_step']['gate_bwd_kernel'])"; done
