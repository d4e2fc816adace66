#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c2/tests.log 2>&1; rc=$?
tail -5 gpurun_out/c2/tests.log
[ $rc -ne 0 ] && exit $rc
BENCH_ARGS="--config c2" ROUNDS=2 bash tools/ab_libs.sh c2_unet pylc_amd/libpylc_hip_r4.so pylc_amd/libpylc_hip.so 2>&1 | tee gpurun_out/c2/ab_unet.txt
BENCH_ARGS="--config c5" ROUNDS=2 bash tools/ab_libs.sh c2_c5 pylc_amd/libpylc_hip_r4.so pylc_amd/libpylc_hip.so 2>&1 | tee gpurun_out/c2/ab_c5.txt
