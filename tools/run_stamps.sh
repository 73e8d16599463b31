#!/bin/bash
# GPU box: diagnostic build with s_memtime stamps in its own copy of the library, phase timings of
# the forward and of k_dhidden_gen; the shipped library is rebuilt afterwards.
set -e
cd $GRAFT_REPO_ROOT/rnnt_amd/csrc
cp librnnt_engine.so /tmp/librnnt_engine.shipped.so
make clean > /dev/null && make -j16 EXTRA=-DRNNT_STAMPS > /tmp/stamps_build.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/exp_stamps.py > gpurun_out/stamps_fwd.log 2>&1 || true
python3 tools/exp_dhgen_stamps.py > gpurun_out/stamps_dhgen.log 2>&1 || true
cd rnnt_amd/csrc && make clean > /dev/null && cp /tmp/librnnt_engine.shipped.so librnnt_engine.so
