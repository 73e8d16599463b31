#!/bin/bash
# A/B of engine libraries on one box at a named bench config: bash tools/ab_cfg.sh <config> <dtype> <lib> [<lib> ...]  ("-" = the shipped library)
cfg=$1; dt=$2; shift 2
for round in 1 2; do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset RNNT_ENGINE_LIB; else export RNNT_ENGINE_LIB=$L; fi
    python3 bench.py --config $cfg --dtype $dt --no-cpu-baseline --steps 6 --warmup 2 --no-parity --no-exact-fp32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg $dt $L', round(d['ms_per_step'], 2), 'ms  loss', d['loss'], {k: round(v, 2) for k, v in d['stages_ms'].items() if v > 0.3})"
  done
done
