#!/bin/bash
# Diagnostic builds of the bf16x3 kernels with parts compiled out (-DX3_EXP=bits, rnnt_amd/csrc/x3.hip): one
# library per variant under build_variants/x3/, every other object taken from the shipped build.
#   tools/build_x3_variants.sh 1 2 4 8 16 32 6 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants/x3
make -C rnnt_amd/csrc -j6 -s librnnt_engine.so
others=$(ls rnnt_amd/csrc/*.o | grep -v x3.o)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DX3_EXP=$v -Irnnt_amd/csrc -c rnnt_amd/csrc/x3.hip -o build_variants/x3/x3_$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/x3/lib_$v.so $others build_variants/x3/x3_$v.o
done
ls build_variants/x3/*.so
