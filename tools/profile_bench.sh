#!/bin/bash
# Profiles the default bench command on the GPU box: kernel-trace stats + HBM traffic counters.
# usage: tools/profile_bench.sh <tag>     (writes gpurun_out/<tag>*)
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- $CMD > gpurun_out/$TAG.trace.log 2>&1
echo "trace rc=$?"
SHORT="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/fetch -- $SHORT > gpurun_out/$TAG.fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/write -- $SHORT > gpurun_out/$TAG.write.log 2>&1
echo "write rc=$?"
grep '"metric"' gpurun_out/$TAG.trace.log | tail -1 > gpurun_out/$TAG.bench.json
