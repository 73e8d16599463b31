#!/bin/bash
# usage: bash tools/r04_run.sh <tag> <cmd...>  — runs under timeout, logs to gpurun_out/<tag>.log, prints the tail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; shift
timeout -k 10 ${LIMIT:-600} "$@" > gpurun_out/$TAG.log 2>&1
rc=$?
tail -${TAIL:-25} gpurun_out/$TAG.log
echo "$TAG rc=$rc"
exit $rc
