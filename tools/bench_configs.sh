#!/bin/bash
# usage: tools/bench_configs.sh <tag> <config> [<config> ...]   — one bench line per config into gpurun_out/<tag>_<config>.json + a summary
TAG=$1; shift
mkdir -p gpurun_out
for c in "$@"; do
  cfg=${c%%:*}; dt=fp32; [[ "$c" == *:bf16 ]] && dt=bf16
  timeout -k 10 300 python3 bench.py --config $cfg --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_${cfg}_${dt}.json 2>gpurun_out/${TAG}_${cfg}_${dt}.err || { echo "$c failed"; tail -3 gpurun_out/${TAG}_${cfg}_${dt}.err; exit 1; }
  python3 - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_${cfg}_${dt}.json"))
print("${cfg} ${dt}", round(j["ms_per_step"], 3), {k: round(v, 3) for k, v in j["stages_ms"].items()}, j.get("parity"))
PY
done
