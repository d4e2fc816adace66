#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c10
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c10/tests.log 2>&1; rc=$?
tail -5 gpurun_out/c10/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_multi.sh 2 "PYLC_NO_FILTER_INTERLEAVE=1" "" 2>&1 | tee gpurun_out/c10/ab_il.txt
BENCH_ARGS="--config c2" bash tools/ab_multi.sh 1 "PYLC_NO_FILTER_INTERLEAVE=1" "" 2>&1 | tee gpurun_out/c10/ab_il_unet.txt
