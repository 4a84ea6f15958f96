# one stream (the whole step as ONE HIP graph) against the two-stream forked tail, same box,
# interleaved pairs.  usage: bash tools/lab/ab_streams.sh [pairs] [batch] [workload] > gpurun_out/ab_streams.txt
N=${1:-3}
B=${2:-65536}
WL=${3:-mmoe_ae30}
run() { python bench.py --workload $WL --batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-lazy --alt-batch 0 --no-configs --no-loss-check $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$WL B=$B $1', 'ms_per_step', d['ms_per_step'], 'Msamples/s', round(d['value']/1e6,2), 'streams', d['config']['streams'], 'dominant_ms', d['roofline']['avg_launch_ms'])"; }
for rep in $(seq 1 $N); do
run two_streams "--streams 2"
run one_stream "--streams 1"
done
