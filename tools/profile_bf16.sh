#!/bin/bash
# Kernel-trace stats of the bf16 bench (GPU box).  usage: tools/profile_bf16.sh <tag>
TAG=${1:-bf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- $CMD > gpurun_out/$TAG.trace.log 2>&1
echo "trace rc=$?"
grep '"metric"' gpurun_out/$TAG.trace.log | tail -1 > gpurun_out/$TAG.bench.json
cat gpurun_out/$TAG/trace/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120 | head -16
