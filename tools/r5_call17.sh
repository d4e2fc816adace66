#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c17
timeout -k 10 600 python -m pytest tests/test_mode3_gpu.py tests/test_ops_gpu.py tests/test_round3_gpu.py -m gpu -q -x > gpurun_out/c17/tests.log 2>&1; rc=$?
tail -3 gpurun_out/c17/tests.log
[ $rc -ne 0 ] && exit $rc
BENCH_ARGS="--config c5" ROUNDS=2 bash tools/ab_libs.sh c17 pylc_amd/libpylc_hip_prev.so pylc_amd/libpylc_hip.so 2>&1 | tee gpurun_out/c17/ab.txt
