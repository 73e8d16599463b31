#!/bin/bash
# A/B of f16x2 library variants on one box: bash tools/ab_x2.sh <lib> [<lib> ...]  ("-" = the shipped library), two rounds each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset RNNT_ENGINE_LIB; else export RNNT_ENGINE_LIB=$L; fi
    python3 bench.py --dtype f16x2 --no-cpu-baseline --steps 10 --no-parity --no-exact-fp32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', round(d['ms_per_step'], 2), 'ms  loss', d['loss'], {k: round(v, 2) for k, v in d['stages_ms'].items() if 'gemm' in k})"
  done
done
