#!/bin/bash
# Ablation builds of the bf16-storage GEMM source: tools/lab/lib_<name>.so = csrc/gemm16.hip compiled with the given -D
# flags, linked with the regular objects of the other sources (MMLREC_LIB=tools/lab/lib_<name>.so selects it).
# usage: build16.sh name "-DG16_TN_WAVES=4"
set -e
cd "$(dirname "$0")/../.."
PKG=$(ls -d mmlrec-a-unified*_amd)
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function $@ \
  -c $PKG/csrc/gemm16.hip -o tools/lab/gemm16_$name.o
objs=$(ls $PKG/lib/obj/*.o | grep -v "/gemm16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/lib_$name.so tools/lab/gemm16_$name.o $objs
rm -f tools/lab/gemm16_$name.o
echo built tools/lab/lib_$name.so
