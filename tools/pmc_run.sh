#!/bin/bash
# usage: tools/pmc_run.sh <tag> -- runs three PMC passes of a short bench (GPU box only)
set -e
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stage-timing --no-parity"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU --output-format csv -d gpurun_out/$TAG/sq -- $CMD > gpurun_out/$TAG.sq.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/tcp -- $CMD > gpurun_out/$TAG.tcp.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/tcc -- $CMD > gpurun_out/$TAG.tcc.log 2>&1
for d in sq tcp tcc; do python3 tools/pmc_summary.py $(ls gpurun_out/$TAG/$d/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.$d.txt; done
cat gpurun_out/$TAG.sq.txt gpurun_out/$TAG.tcp.txt gpurun_out/$TAG.tcc.txt
