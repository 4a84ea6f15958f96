run() { env $1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'], round(d['value']/1e6,2), d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
run "X=0"
run "MMLREC_GATHER_WGMAX=0"
run "MMLREC_GEMM_WS=0"
run "MMLREC_AMAX_MERGE=0 MMLREC_GATHER_WGMAX=0"
done
