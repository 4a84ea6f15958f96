#!/bin/bash
# builds lib/lab/libmmlrec_wg.so = the product's objects + the weight-gradient experiment (tools/lab/experiments/gemm_wgws.hip)
set -e
cd "$(dirname "$0")/../.."
P=mmlrec-a-unified-multi-task-and-multi-scenario-learning-benchmark-for-recommendation_amd
mkdir -p $P/lib/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$P/csrc $WG_EXTRA -c tools/lab/experiments/gemm_wgws.hip -o $P/lib/lab/gemm_wgws.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/lab/libmmlrec_wg.so $P/lib/obj/*.o $P/lib/lab/gemm_wgws.o
echo built $P/lib/lab/libmmlrec_wg.so
