for rep in 1 2 3; do for w in 1 0; do
MMLREC_GEMM_WS=$w python bench.py --workload pepnet_amazon --steps 40 --warmup 8 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pep ws=$w', d['ms_per_step'], round(d['value']/1e6,2))"
done; done
for rep in 1 2; do for w in 1 0; do
MMLREC_GEMM_WS=$w python bench.py --workload ple_ijcai --steps 40 --warmup 8 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ple ws=$w', d['ms_per_step'], round(d['value']/1e6,2))"
done; done
