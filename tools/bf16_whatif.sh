#!/bin/bash
# Removal tables of the bf16 route's three GEMM kernels (round 6): one diagnostic library per variant under build_variants/bf16w/ —
# bf16.hip with -DBF_CLOCK (in-kernel clock stamps of workgroup 0) and ONE of -DFR_EXP / -DDH_EXP / -DBW_EXP = bits (parts of
# k_joint_fwd_bf16_ra / k_dhidden_bf16 / k_dw_bf16 compiled out: results wrong by construction, cost right), engine.hip with -DRNNT_STAMPS
# (so that the stamp buffer reaches the kernels), every other object from the shipped build.
#   tools/bf16_whatif.sh build                       (build container)
#   tools/bf16_whatif.sh run > profiles/...txt       (GPU box: wall clock of each kernel + the clock it ran at, per variant)
set -e
cd "$(dirname "$0")/.."
D=build_variants/bf16w
FR="${FR_LIST:-0 1 2 4 8 16 32 6 63}"
DH="${DH_LIST:-0 1 2 4 8 16 32 12 63}"
BW="${BW_LIST:-0 1 2 4 8 16 32 64 29}"
if [ "$1" = build ]; then
  mkdir -p $D
  make -C rnnt_amd/csrc -j6 -s librnnt_engine.so
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Irnnt_amd/csrc"
  /opt/rocm/bin/hipcc $F -DRNNT_STAMPS -c rnnt_amd/csrc/engine.hip -o $D/engine_stamps.o &
  n=0
  for spec in $(for v in $FR; do echo FR_EXP:$v; done) $(for v in $DH; do echo DH_EXP:$v; done) $(for v in $BW; do echo BW_EXP:$v; done); do
    var=${spec%%:*}; v=${spec##*:}
    if [ "$v" = 0 ] && [ "$var" != FR_EXP ]; then continue; fi   # one baseline library serves all three kernels
    /opt/rocm/bin/hipcc $F -DBF_CLOCK -D$var=$v -c rnnt_amd/csrc/bf16.hip -o $D/bf16_${var}_$v.o &
    n=$((n+1)); if [ $((n % 6)) = 0 ]; then wait; fi
  done
  wait
  others=$(ls rnnt_amd/csrc/*.o | grep -v -E "/(bf16|engine)\.o")
  for o in $D/bf16_*.o; do
    b=$(basename $o .o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_${b#bf16_}.so $others $D/engine_stamps.o $o
  done
  ls $D/*.so | wc -l
  exit 0
fi
# run: baseline first and last (box drift), each variant once
echo "# bf16 route, cfg2 (B=32,T=1000,U=200,H=512,V=1024): kernel wall clock (HIP events, median of 5) and in-kernel clock of workgroup 0"
echo "# variant = parts compiled out (results wrong by construction).  fwd = k_joint_fwd_bf16_ra<8>, dh = k_dhidden_bf16<true>, dw = k_dw_bf16"
python3 tools/exp_bf16_clock.py FR_EXP_0:all
for v in $FR; do [ $v = 0 ] || python3 tools/exp_bf16_clock.py FR_EXP_$v:fwd; done
for v in $DH; do [ $v = 0 ] || python3 tools/exp_bf16_clock.py DH_EXP_$v:dh; done
for v in $BW; do [ $v = 0 ] || python3 tools/exp_bf16_clock.py BW_EXP_$v:dw; done
python3 tools/exp_bf16_clock.py FR_EXP_0:all
