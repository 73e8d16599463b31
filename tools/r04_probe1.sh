#!/bin/bash
# round 4, first GPU call: MFMA shape probe, cfg4 on the bf16x3 route, bf16x3 shard timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 ./tools/mfma_shape 3000 7 > gpurun_out/r04.mfma_shape.txt 2>&1; echo "mfma_shape rc=$?"
cat gpurun_out/r04.mfma_shape.txt
timeout -k 10 300 python3 bench.py --config cfg4 --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04.cfg4_bf16x3.json 2> gpurun_out/r04.cfg4_bf16x3.err; echo "cfg4 rc=$?"
timeout -k 10 300 python3 tools/exp_batch_scaling.py bf16x3 > gpurun_out/r04.batch_scaling_bf16x3.txt 2>&1; echo "scaling rc=$?"
cat gpurun_out/r04.batch_scaling_bf16x3.txt
