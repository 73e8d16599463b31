#!/bin/bash
# Diagnostic library with the k-step stamps of k_joint_fwd_x2 (-DRNNT_STAMPS): build_variants/x3/lib_x2_stamps.so
#   STAMP_DTYPE=f16x2 RNNT_ENGINE_LIB=build_variants/x3/lib_x2_stamps.so python3 tools/exp_x3_stamps.py 1
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants/x3
make -C rnnt_amd/csrc -j6 -s librnnt_engine.so
others=$(ls rnnt_amd/csrc/*.o | grep -v -E "/(x2|engine)\.o")
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DRNNT_STAMPS -Irnnt_amd/csrc"
/opt/rocm/bin/hipcc $F -c rnnt_amd/csrc/engine.hip -o build_variants/x3/engine_stamps.o &
/opt/rocm/bin/hipcc $F "$@" -c rnnt_amd/csrc/x2.hip -o build_variants/x3/x2_stamps.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/x3/lib_x2_stamps.so $others build_variants/x3/engine_stamps.o build_variants/x3/x2_stamps.o
ls -la build_variants/x3/lib_x2_stamps.so
