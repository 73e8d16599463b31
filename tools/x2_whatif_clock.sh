#!/bin/bash
# Round 6: the main classes of round 5's removal tables of k_joint_fwd_x2 / k_dhidden_x2 (profiles/r05_fwd_whatif.txt, r05_dhidden_whatif.txt) once
# more WITH the in-kernel clock of workgroup 0 (-DRNNT_STAMPS: X2_CLOCK_STAMP) beside every wall clock — are the f16x2 kernels power-bound like the
# bf16 route's (profiles/r06_bf16_whatif.txt)?  One diagnostic library per variant under build_variants/x2c/.
#   tools/x2_whatif_clock.sh build            (build container)
#   tools/x2_whatif_clock.sh run              (GPU box)
set -e
cd "$(dirname "$0")/.."
D=build_variants/x2c
FW="0 128 2 2048 256 32"     # forward: nothing | MFMAs | W's bytes | fragment reads | logits stores | production arithmetic
DH="1 2 24 32"               # dHidden: MFMAs | W's bytes | logits reads + G stores | fragment reads
if [ "$1" = build ]; then
  mkdir -p $D
  make -C rnnt_amd/csrc -j6 -s librnnt_engine.so
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Irnnt_amd/csrc -DRNNT_STAMPS"
  /opt/rocm/bin/hipcc $F -c rnnt_amd/csrc/engine.hip -o $D/engine_stamps.o &
  for v in $FW; do /opt/rocm/bin/hipcc $F -DX2_EXP=$v -c rnnt_amd/csrc/x2.hip -o $D/x2_F$v.o & done
  wait
  for v in $DH; do /opt/rocm/bin/hipcc $F -DXG2_EXP=$v -c rnnt_amd/csrc/x2.hip -o $D/x2_G$v.o & done
  wait
  others=$(ls rnnt_amd/csrc/*.o | grep -v -E "/(x2|engine)\.o")
  for o in $D/x2_*.o; do b=$(basename $o .o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_${b#x2_}.so $others $D/engine_stamps.o $o; done
  rm -f $D/*.o; ls $D
  exit 0
fi
echo "# f16x2 route, cfg2: kernel wall clock (HIP events, median of 5) and the in-kernel clock of workgroup 0; variant = part compiled out (results wrong by construction)"
python3 tools/exp_x2_clock.py F0:all
for v in $FW; do [ $v = 0 ] || python3 tools/exp_x2_clock.py F$v:fwd; done
for v in $DH; do python3 tools/exp_x2_clock.py G$v:dh; done
python3 tools/exp_x2_clock.py F0:all
