#!/bin/bash
# GPU box, round 6, the f16x2 default route: everything profiles/r06_* is built from.
# usage: bash tools/profile_r06.sh [part]
#   part 1: bench lines (default = f16x2 with exact_fp32 + bf16x3 + CPU baseline; cfg4 / cfg5 / ref1024 / permuted enc on f16x2)
#   part 2: rocprofv3 kernel stats (f16x2) + FETCH_SIZE / WRITE_SIZE passes
#   part 3: SQ counters of the f16x2 kernels; f16x2 shard timings (B = 32/16/8/4)
PART=${1:-all}
TAG=r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
  timeout -k 10 500 python3 bench.py > gpurun_out/$TAG.default.json 2> gpurun_out/$TAG.default.err; echo "default rc=$?"
  timeout -k 10 200 python3 bench.py --permuted-enc --no-cpu-baseline --no-exact-fp32 > gpurun_out/$TAG.f16x2.permuted.json 2>/dev/null; echo "permuted rc=$?"
  timeout -k 10 200 python3 bench.py --config ref1024 --no-cpu-baseline --no-exact-fp32 > gpurun_out/$TAG.ref1024.f16x2.json 2>/dev/null; echo "ref1024 rc=$?"
  timeout -k 10 300 python3 bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 > gpurun_out/$TAG.cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
  timeout -k 10 300 python3 bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 > gpurun_out/$TAG.cfg4.json 2>/dev/null; echo "cfg4 rc=$?"
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
  DT=f16x2
  CMD="python3 bench.py --dtype $DT --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-exact-fp32"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG.$DT/trace -- $CMD > gpurun_out/$TAG.$DT.trace.log 2>&1
  echo "$DT trace rc=$?"
  grep '"metric"' gpurun_out/$TAG.$DT.trace.log | tail -1 > gpurun_out/$TAG.$DT.under_rocprof.json
  cp $(ls gpurun_out/$TAG.$DT/trace/*/*kernel_stats.csv | head -1) gpurun_out/$TAG.$DT.kernel_stats.csv
  SHORT="python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing --no-exact-fp32"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$TAG.$DT/$C -- $SHORT > gpurun_out/$TAG.$DT.$C.log 2>&1
    echo "$DT $C rc=$?"
    python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.$DT/$C/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.$DT.$C.txt
  done
fi
if [ "$PART" = 3 ] || [ "$PART" = all ]; then
  DT=f16x2
  SHORT="python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing --no-exact-fp32"
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG.$DT/SQ -- $SHORT > gpurun_out/$TAG.$DT.SQ.log 2>&1
  echo "$DT SQ rc=$?"
  python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.$DT/SQ/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.$DT.SQ.txt
  timeout -k 10 300 python3 tools/exp_batch_scaling.py f16x2 > gpurun_out/$TAG.batch_scaling_f16x2.txt 2>&1; echo "scaling rc=$?"
fi
if [ "$PART" = 4 ] || [ "$PART" = all ]; then
  # the other routes' lines (bf16 = BASELINE config 3's arithmetic), the projections, shard timings on bf16
  timeout -k 10 200 python3 bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/$TAG.bf16.json 2>/dev/null; echo "bf16 rc=$?"
  timeout -k 10 300 python3 tools/bench_linear.py > gpurun_out/$TAG.linear.txt 2>&1; echo "linear rc=$?"
  timeout -k 10 300 python3 tools/exp_batch_scaling.py bf16 > gpurun_out/$TAG.batch_scaling_bf16.txt 2>&1; echo "bf16 scaling rc=$?"
  timeout -k 10 300 python3 tools/bench_decode.py 1.6 1.9 > gpurun_out/$TAG.bench_decode.txt 2>&1; echo "decode rc=$?"
  # two self-launched ranks on the one GPU (gloo all-reduce): the launch / file rendezvous / sharding / barrier / max-over-ranks path of `bench.py --gpus 2`
  BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu-baseline --no-exact-fp32 > gpurun_out/$TAG.two_rank.json 2> gpurun_out/$TAG.two_rank.err; echo "two-rank rc=$?"
fi
