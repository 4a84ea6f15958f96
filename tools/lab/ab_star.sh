for rep in 1 2 3; do
for pl in 1 0; do
MMLREC_STAR_PLANES=$pl python bench.py --workload star_amazon --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('star_planes=$pl', d['ms_per_step'], round(d['value']/1e6,2), {k:v for k,v in d['kernels_ms_per_step'].items() if 'gemm' in k})"
done; done
