#!/bin/bash
# Lab builds of the activation-stationary forward kernel (csrc/gemm_panel.hip) with parts of the k-step switched off at
# COMPILE time (-DPN_LAB=n: 1 no epilogue slices, 2 no weight DMA / waits, 8 no MFMAs, 16 panel loaded once; results are
# garbage): lib/lab/libmmlrec_panel_lab<n>.so, linked from the product's other objects.  Run on the GPU box:
#   tools/lab/panel_lab.sh 1 2 3 19 27 ; for n in ...; do MMLREC_LIB=.../libmmlrec_panel_lab$n.so python tools/bench_gemm.py; done
set -e
cd "$(dirname "$0")/../.."
P=mmlrec-a-unified-multi-task-and-multi-scenario-learning-benchmark-for-recommendation_amd
mkdir -p $P/lib/lab
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DPN_LAB=$n $PN_EXTRA -c $P/csrc/gemm_panel.hip -o $P/lib/lab/gemm_panel_lab$n.o
  objs=$(ls $P/lib/obj/*.o | grep -v gemm_panel.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/lab/libmmlrec_panel_lab$n.so $objs $P/lib/lab/gemm_panel_lab$n.o
  echo built $P/lib/lab/libmmlrec_panel_lab$n.so
done
