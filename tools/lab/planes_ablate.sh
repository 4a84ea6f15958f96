#!/bin/bash
mkdir -p gpurun_out
for v in "" ${QVARIANTS:-q_NO_MFMA q_NO_DMA}; do
  echo "== ${v:-full}"
  if [ -n "$v" ]; then export MMLREC_LIB=$PWD/tools/lab/lib_$v.so; else unset MMLREC_LIB; fi
  timeout 200 python tools/lab/bench_planes.py 2>&1 | grep -v amdgpu.ids
done
