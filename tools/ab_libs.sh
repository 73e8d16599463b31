#!/bin/bash
# A/B of engine builds on one GPU box: tools/ab_libs.sh <bench args> -- lib1.so lib2.so ...
# (libraries under build_variants/, built with make EXTRA=...); prints ms/step and the stage times.
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
cp rnnt_amd/csrc/librnnt_engine.so /tmp/_orig.so
for rep in 1 2; do
for lib in "$@"; do
  cp "$lib" rnnt_amd/csrc/librnnt_engine.so
  python bench.py "${args[@]}" --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(j['ms_per_step'],3), {k: round(v,2) for k,v in j['stages_ms'].items() if v > 0.5})"
done; done
cp /tmp/_orig.so rnnt_amd/csrc/librnnt_engine.so
