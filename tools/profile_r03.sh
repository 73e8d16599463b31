#!/bin/bash
# GPU box, round 3: everything profiles/r03_* is built from.  usage: bash tools/profile_r03.sh [part]
#   part 1: bench lines (default = bf16x3 with CPU baseline, fp32, bf16, permuted enc, ref1024 x 3 dtypes)
#   part 2: rocprofv3 kernel stats (bf16x3, fp32, bf16) + FETCH_SIZE / WRITE_SIZE passes (bf16x3, bf16)
#   part 3: SQ counters of the bf16x3 and bf16 kernels; kernel stats of the (f) rows (predictor, optim, decode)
PART=${1:-all}
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
  timeout -k 10 400 python3 bench.py > gpurun_out/$TAG.default.json 2> gpurun_out/$TAG.default.err; echo "default rc=$?"
  timeout -k 10 200 python3 bench.py --dtype fp32 --no-cpu-baseline > gpurun_out/$TAG.fp32.json 2>/dev/null; echo "fp32 rc=$?"
  timeout -k 10 200 python3 bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/$TAG.bf16.json 2>/dev/null; echo "bf16 rc=$?"
  for DT in bf16x3 fp32; do
    timeout -k 10 200 python3 bench.py --dtype $DT --permuted-enc --no-cpu-baseline > gpurun_out/$TAG.$DT.permuted.json 2>/dev/null; echo "$DT permuted rc=$?"
  done
  for DT in bf16x3 fp32 bf16; do
    timeout -k 10 200 python3 bench.py --config ref1024 --dtype $DT --no-cpu-baseline > gpurun_out/$TAG.ref1024.$DT.json 2>/dev/null
    timeout -k 10 200 python3 bench.py --config ref1024 --dtype $DT --permuted-enc --no-cpu-baseline > gpurun_out/$TAG.ref1024.$DT.permuted.json 2>/dev/null
    echo "ref1024 $DT rc=$?"
  done
  timeout -k 10 300 python3 bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG.cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
  for DT in bf16x3 fp32 bf16; do
    CMD="python3 bench.py --dtype $DT --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG.$DT/trace -- $CMD > gpurun_out/$TAG.$DT.trace.log 2>&1
    echo "$DT trace rc=$?"
    grep '"metric"' gpurun_out/$TAG.$DT.trace.log | tail -1 > gpurun_out/$TAG.$DT.under_rocprof.json
    cp $(ls gpurun_out/$TAG.$DT/trace/*/*kernel_stats.csv | head -1) gpurun_out/$TAG.$DT.kernel_stats.csv
  done
  for DT in bf16x3 bf16; do
    SHORT="python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing"
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$TAG.$DT/$C -- $SHORT > gpurun_out/$TAG.$DT.$C.log 2>&1
      echo "$DT $C rc=$?"
      python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.$DT/$C/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.$DT.$C.txt
    done
  done
fi
if [ "$PART" = 3 ] || [ "$PART" = all ]; then
  for DT in bf16x3 bf16; do
    SHORT="python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing"
    timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG.$DT/SQ -- $SHORT > gpurun_out/$TAG.$DT.SQ.log 2>&1
    echo "$DT SQ rc=$?"
    python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.$DT/SQ/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.$DT.SQ.txt
  done
  timeout -k 10 200 python3 tools/bench_predictor.py > gpurun_out/$TAG.bench_predictor.txt 2>&1
  timeout -k 10 200 python3 tools/bench_optim.py > gpurun_out/$TAG.bench_optim.json 2>/dev/null
  timeout -k 10 200 python3 tools/bench_decode.py > gpurun_out/$TAG.bench_decode.txt 2>&1
  for T in predictor optim decode; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG.f_$T/trace -- python3 tools/bench_$T.py > gpurun_out/$TAG.f_$T.trace.log 2>&1
    cp $(ls gpurun_out/$TAG.f_$T/trace/*/*kernel_stats.csv | head -1) gpurun_out/$TAG.f_$T.kernel_stats.csv
    echo "f_$T trace rc=$?"
  done
fi
