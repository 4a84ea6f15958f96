# Weight-gradient kernel choice in the real step, interleaved on one box: MMLREC_GEMM_NT = 0 (tile kernel), 2 (default:
# cut-once kernel for launches of >= 16 tiles), 1 (every qualifying launch).  usage: bash tools/lab/ab_nt.sh [reps] [workload]
N=${1:-2}
WL=${2:-mmoe_ae30}
run() { MMLREC_GEMM_NT=$1 python bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms_per_step']; print('$WL NT=$1', 'ms_per_step', d['ms_per_step'], 'Msamples/s', round(d['value']/1e6,2), {a[:40]: b for a, b in k.items() if 'gemm_nt' in a or 'false, false' in a or 'slab_reduce' in a})"; }
for rep in $(seq 1 $N); do
run 0
run 2
run 1
done
