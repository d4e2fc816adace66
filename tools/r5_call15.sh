#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c15
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/c15/tests.log 2>&1; rc=$?
tail -4 gpurun_out/c15/tests.log
[ $rc -ne 0 ] && exit $rc
BENCH_ARGS="--config c5" bash tools/ab_multi.sh 2 "PYLC_DW_WGRAD_MAIN=1" "" 2>&1 | tee gpurun_out/c15/ab_c5.txt
