#!/bin/bash
# builds lib/lab/libmmlrec_wg.so = the product's objects + the weight-gradient experiment (tools/lab/experiments/gemm_wgws.hip)
set -e
cd "$(dirname "$0")/../.."
P=mmlrec-a-unified-multi-task-and-multi-scenario-learning-benchmark-for-recommendation_amd
mkdir -p $P/lib/lab
sed 's|#include "common.hpp"|#include "../../../'$P'/csrc/common.hpp"|; s|#include "lds_async.hpp"|#include "../../../'$P'/csrc/lds_async.hpp"|; s|#include "reduce.hpp"|#include "../../../'$P'/csrc/reduce.hpp"|' tools/lab/experiments/gemm_wgws.hip > /tmp/gemm_wgws_lab.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$P/csrc -I. $WG_EXTRA -c /tmp/gemm_wgws_lab.hip -o $P/lib/lab/gemm_wgws.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/lab/libmmlrec_wg.so $P/lib/obj/*.o $P/lib/lab/gemm_wgws.o
echo built $P/lib/lab/libmmlrec_wg.so
