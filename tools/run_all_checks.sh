#!/bin/bash
# GPU box: everything that guards the kernels beyond `pytest -m gpu`, in the order it should be run after a
# kernel change.  STOPS AT THE FIRST FAILURE: every check writes its own log under gpurun_out/checks/, the
# exit status tested is the python process's own (never a pipe's last stage), and the seeds of the
# randomised sweeps are printed before each run so a failing sweep can be replayed.  A memory fault ends
# the run — read the log, do not loop over it.
#   gpurun --timeout 1200 -- 'bash tools/run_all_checks.sh > gpurun_out/checks.log 2>&1'
set -u -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/checks
fails=0
run() {  # run <name> <timeout s> <cmd...>
    local name=$1 limit=$2; shift 2
    local log=gpurun_out/checks/$name.log
    echo "== $name: $*"
    timeout -k 10 "$limit" "$@" > "$log" 2>&1
    local rc=$?
    tail -3 "$log"
    if [ $rc -ne 0 ]; then
        echo "FAILED: $name (exit $rc; 124/137 = timeout kill, 134/139 = abort/fault) — see $log"
        exit $rc
    fi
}
S1=${SEED1:-$RANDOM}; S2=${SEED2:-$RANDOM}; S3=${SEED3:-$RANDOM}; S4=${SEED4:-$RANDOM}; S5=${SEED5:-$RANDOM}
echo "seeds: fuzz_parity=$S1 fuzz_lattice=$S2 fuzz_misc=$S3 fuzz_blank=$S4 fuzz_pad=$S5 (replay: SEED1=.. SEED5=.. bash tools/run_all_checks.sh)"
run pytest_gpu 900 python3 -m pytest tests -q -x -m gpu
run fuzz_parity 900 python3 tools/fuzz_parity.py 150 60 "$S1" 400 100 120
run fuzz_lattice 300 python3 tools/fuzz_lattice.py 40 20 "$S2"
run fuzz_misc 300 python3 tools/fuzz_misc.py 30 "$S3"
run fuzz_blank 300 python3 tools/fuzz_blank.py 60 "$S4"
run fuzz_pad 300 python3 tools/fuzz_pad.py 60 "$S5"
[ -f tools/libguard.so ] || /opt/rocm/bin/hipcc -shared -fPIC -o tools/libguard.so tools/guard_alloc.hip || exit 1
run guard_sweep 300 python3 tools/guard_sweep.py
echo "ALL CHECKS PASSED"
