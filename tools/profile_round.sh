#!/bin/bash
# GPU box: everything profiles/ is built from — default bench line, rocprofv3 kernel stats and the
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes) for the fp32 and bf16 routes.
# usage: tools/profile_round.sh <tag>      (writes gpurun_out/<tag>.*, then tools/collect_profiles.py)
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/$TAG.default.json 2> gpurun_out/$TAG.default.err
echo "default rc=$?"
for DT in fp32 bf16; do
  CMD="python3 bench.py --dtype $DT --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG.$DT/trace -- $CMD > gpurun_out/$TAG.$DT.trace.log 2>&1
  echo "$DT trace rc=$?"
  grep '"metric"' gpurun_out/$TAG.$DT.trace.log | tail -1 > gpurun_out/$TAG.$DT.under_rocprof.json
  cp $(ls gpurun_out/$TAG.$DT/trace/*/*kernel_stats.csv | head -1) gpurun_out/$TAG.$DT.kernel_stats.csv
  SHORT="python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$TAG.$DT/$C -- $SHORT > gpurun_out/$TAG.$DT.$C.log 2>&1
    echo "$DT $C rc=$?"
    python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.$DT/$C/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.$DT.$C.txt
  done
done
timeout -k 10 300 python3 bench.py --dtype bf16 > gpurun_out/$TAG.bf16.default.json 2>/dev/null
echo "bf16 default rc=$?"
