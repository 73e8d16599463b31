#!/bin/bash
# Round 6's robustness record (profiles/r06_robustness.txt), on the GPU box: randomised parity, guard pages, soak, the lattice / blank / padding
# fuzzers, the decode fuzzer, the counted-vmcnt check.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_robustness.txt
{
echo "# tools/robustness_r06.sh on the round's final code (one gpurun call)"
echo "# tools/fuzz_parity.py 0 40 6066 400 512 0 300   (300 random ragged f16x2 cases, any H, V up to 1600 / 2048, poisoned workspaces; 40 bf16 cases)"
timeout -k 10 500 python3 tools/fuzz_parity.py 0 40 6066 400 512 0 300 2>&1 | grep -v amdgpu.ids | tail -1
echo "# tools/guard_sweep.py   (every input ending at an unmapped page: fused shapes x 3 routes, unfused entries, projections, ConvPredictor, decode scan, optimizer)"
timeout -k 10 300 python3 tools/guard_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "# soak cfg2 60 s f16x2"
timeout -k 10 120 python3 tools/soak.py cfg2 60 f16x2 2>&1 | grep -v amdgpu.ids | tail -1
echo "# soak cfg4 25 s f16x2   (k_dw_x2m: whole-block + tall dW tiles)"
timeout -k 10 120 python3 tools/soak.py cfg4 25 f16x2 2>&1 | grep -v amdgpu.ids | tail -1
echo "# soak cfg2 40 s bf16"
timeout -k 10 100 python3 tools/soak.py cfg2 40 bf16 2>&1 | grep -v amdgpu.ids | tail -1
for f in fuzz_lattice fuzz_misc fuzz_blank fuzz_pad; do echo "# $f"; timeout -k 10 300 python3 tools/$f.py 2>&1 | grep -v amdgpu.ids | tail -1; done
echo "# tools/fuzz_decode.py 80 11   (persistent launch = kernel-per-layer loop = per-frame loop = the numpy oracle)"
timeout -k 10 300 python3 tools/fuzz_decode.py 80 11 2>&1 | grep -v amdgpu.ids | tail -1
} > $O 2>&1
cat $O
grep -q "Memory access fault" $O && exit 1
exit 0
