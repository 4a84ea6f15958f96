run() { MMLREC_DEFER_REDUCE=$1 MMLREC_GATHER_WGMAX=$2 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('defer=$1 wgmax=$2', d['ms_per_step'], round(d['value']/1e6,2))"; }
for i in 1 2 3; do run 0 0; run 1 0; run 0 1; done
