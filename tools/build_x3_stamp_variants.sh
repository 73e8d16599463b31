#!/bin/bash
# Diagnostic libraries with the in-kernel clock stamps of k_joint_fwd_x3 (-DRNNT_STAMPS) for X3_EXP variants:
#   tools/build_x3_stamp_variants.sh 0 138 ...   -> build_variants/x3/lib_stamps_<exp>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants/x3
make -C rnnt_amd/csrc -j6 -s librnnt_engine.so
others=$(ls rnnt_amd/csrc/*.o | grep -v -E "/(x3|engine)\.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DRNNT_STAMPS -Irnnt_amd/csrc -c rnnt_amd/csrc/engine.hip -o build_variants/x3/engine_stamps.o
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DRNNT_STAMPS -DX3_EXP=$v -Irnnt_amd/csrc -c rnnt_amd/csrc/x3.hip -o build_variants/x3/x3_stamps_$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/x3/lib_stamps_$v.so $others build_variants/x3/engine_stamps.o build_variants/x3/x3_stamps_$v.o
done
ls build_variants/x3/lib_stamps_*.so
