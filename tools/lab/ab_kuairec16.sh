# KuaiRec-32 bf16 (GEMM mode 1 + bf16 storage), interleaved A/B of library builds.  usage: ab_kuairec16.sh lib1 lib2 ...
run() { MMLREC_LIB=$2 MMLREC_GEMM_MODE=1 python bench.py --workload mmoe_kuairec --no-configs --no-cpu-baseline --no-lazy --alt-batch 0 --table-update auto --steps 30 --warmup 5 --no-loss-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms_per_step']; print('$1', d['ms_per_step'], round(d['value']/1e6,2), {a[:34]: b for a, b in list(k.items())[:9]})"; }
for rep in 1 2; do
for lib in "$@"; do run $(basename $lib) $lib; done
done
