#!/bin/bash
# Diagnostic builds of the bf16 route's forward (-DFR_EXP=bits) or dHidden kernel (VAR=DH_EXP) with parts compiled out: one
# library per variant under build_variants/bf16/, every other object taken from the shipped build.
#   tools/build_bf16_variants.sh 1 2 4 8 16 32 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants/bf16
make -C rnnt_amd/csrc -j6 -s librnnt_engine.so
others=$(ls rnnt_amd/csrc/*.o | grep -v bf16.o)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -D${VAR:-FR_EXP}=$v $FR_EXTRA -Irnnt_amd/csrc -c rnnt_amd/csrc/bf16.hip -o build_variants/bf16/bf16_${VAR:-FR_EXP}_$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/bf16/lib_${VAR:-FR_EXP}_$v.so $others build_variants/bf16/bf16_${VAR:-FR_EXP}_$v.o
done
ls build_variants/bf16/*.so
