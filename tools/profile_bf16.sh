#!/bin/bash
# Profiles the bf16 bench (GPU box): kernel-trace stats, then HBM traffic counters in separate passes.
# usage: tools/profile_bf16.sh <tag> [pmc]
TAG=${1:-bf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- $CMD > gpurun_out/$TAG.trace.log 2>&1
echo "trace rc=$?"
grep '"metric"' gpurun_out/$TAG.trace.log | tail -1 > gpurun_out/$TAG.bench.json
cat gpurun_out/$TAG/trace/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120 | head -8
if [ "$2" = "pmc" ]; then
  SHORT="python3 bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing"
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/fetch -- $SHORT > gpurun_out/$TAG.fetch.log 2>&1
  echo "fetch rc=$?"
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/write -- $SHORT > gpurun_out/$TAG.write.log 2>&1
  echo "write rc=$?"
  python3 tools/pmc_summary.py $(ls gpurun_out/$TAG/fetch/*/*counter_collection.csv | head -1)
  python3 tools/pmc_summary.py $(ls gpurun_out/$TAG/write/*/*counter_collection.csv | head -1)
fi
