#!/bin/bash
# Ablation builds of the GEMM source: tools/lab/lib_<name>.so = csrc/gemm.hip compiled with the given -D flags, linked
# with the regular objects of the other sources.  usage: build.sh name "-DMML_LAB_BN=128 -DMML_LAB_EMU=3 -D..." 
set -e
cd "$(dirname "$0")/../.."
PKG=$(ls -d mmlrec-a-unified*_amd)
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -DMML_LAB $@ \
  -c $PKG/csrc/gemm.hip -o tools/lab/gemm_$name.o
objs=$(ls $PKG/lib/obj/*.o | grep -v gemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/lib_$name.so tools/lab/gemm_$name.o $objs
rm -f tools/lab/gemm_$name.o
echo built tools/lab/lib_$name.so
