#!/bin/bash
# GPU box: everything that guards the kernels beyond `pytest -m gpu`, in the order it should be run after a
# kernel change (stops at the first failure; a memory fault ends the run — do not loop over it).
#   gpurun --timeout 1200 -- 'bash tools/run_all_checks.sh > gpurun_out/checks.log 2>&1; tail -20 gpurun_out/checks.log'
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest -m gpu";            timeout -k 10 900 python3 -m pytest tests -q -x -m gpu 2>&1 | tail -2
echo "== fuzz: fused path, poisoned workspaces (fp32 + bf16)"; timeout -k 10 600 python3 tools/fuzz_parity.py 150 60 $RANDOM 400 100 | tail -1
echo "== fuzz: lattice";            timeout -k 10 300 python3 tools/fuzz_lattice.py 40 20 $RANDOM | tail -1
echo "== fuzz: other entry points"; timeout -k 10 300 python3 tools/fuzz_misc.py 30 $RANDOM | tail -1
echo "== fuzz: blank index";        timeout -k 10 300 python3 tools/fuzz_blank.py 60 $RANDOM | tail -1
echo "== fuzz: padded H / V";       timeout -k 10 300 python3 tools/fuzz_pad.py 60 $RANDOM | tail -1
echo "== inputs against unmapped pages"
[ -f tools/libguard.so ] || /opt/rocm/bin/hipcc -shared -fPIC -o tools/libguard.so tools/guard_alloc.hip
timeout -k 10 300 python3 tools/guard_sweep.py | grep -v "unmapped from" | tail -3
echo "ALL CHECKS PASSED"
