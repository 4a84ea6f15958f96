for f in 0 1; do MMLREC_PEP_FUSE=$f python bench.py --workload pepnet_amazon --steps 40 --warmup 8 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fuse=$f', d['ms_per_step'], round(d['value']/1e6,2)); [print('   ',k,v) for k,v in d['kernels_ms_per_step'].items()]"; done
python bench.py --workload ple_ijcai --steps 40 --warmup 8 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ple', d['ms_per_step'], round(d['value']/1e6,2)); [print('   ',k,v) for k,v in d['kernels_ms_per_step'].items()]"
