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
timeout -k 10 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 > gpurun_out/$TAG.bf16.default.json 2>/dev/null
echo "bf16 default rc=$?"
# MFMA utilisation of the SHIPPED kernels (north_star: "MFMA utilisation on the projection"): SQ counters,
# their own pass (8 SQ slots), fp32 route
SHORT="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stage-timing"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG.fp32/SQ -- $SHORT > gpurun_out/$TAG.fp32.SQ.log 2>&1
echo "fp32 SQ rc=$?"
python3 tools/pmc_summary.py $(ls gpurun_out/$TAG.fp32/SQ/*/*counter_collection.csv | head -1) > gpurun_out/$TAG.fp32.SQ.txt
# the reference's joint width (H = 1024) and the long-utterance config: bench lines + kernel stats
for C in ref1024 cfg4; do
  timeout -k 10 300 python3 bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG.$C.json 2>/dev/null
  echo "$C rc=$?"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG.ref1024/trace -- python3 bench.py --config ref1024 --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/$TAG.ref1024.trace.log 2>&1
cp $(ls gpurun_out/$TAG.ref1024/trace/*/*kernel_stats.csv | head -1) gpurun_out/$TAG.ref1024.kernel_stats.csv
timeout -k 10 300 python3 bench.py --config ref1024 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG.ref1024.bf16.json 2>/dev/null
echo "ref1024 bf16 rc=$?"
