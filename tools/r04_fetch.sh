#!/bin/bash
# FETCH_SIZE of one f16x2 step (rocprofv3 --pmc): bash tools/r04_fetch.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-fetch}
SHORT="python3 bench.py --dtype f16x2 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing --no-exact-fp32"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG.pmc -- $SHORT > gpurun_out/$TAG.pmc.log 2>&1
echo "FETCH rc=$?"
python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.pmc/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.txt
cat gpurun_out/$TAG.txt
