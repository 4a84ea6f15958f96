#!/bin/bash
# Ablation build of ONE source: tools/lab/lib_<name>.so = csrc/<src>.hip compiled with the given -D flags, linked with
# the regular objects of the other sources.  usage: build_src.sh gather_scatter noflush "-DMML_LAB_SC_NOFLUSH"
set -e
cd "$(dirname "$0")/../.."
PKG=$(ls -d mmlrec-a-unified*_amd)
src=$1; name=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -DMML_LAB $@ \
  -c $PKG/csrc/$src.hip -o tools/lab/${src}_$name.o
objs=$(ls $PKG/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/lib_$name.so tools/lab/${src}_$name.o $objs
rm -f tools/lab/${src}_$name.o
echo built tools/lab/lib_$name.so
