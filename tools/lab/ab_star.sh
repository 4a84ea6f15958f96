for rep in 1 2; do
for pl in 1 0; do
MMLREC_GEMM_PLANES=$pl python bench.py --workload star_amazon --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('planes=$pl', d['ms_per_step'], d['value'])"
done; done
