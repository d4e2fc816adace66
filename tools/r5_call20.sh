#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c20
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_fullsize_ops_gpu.py tests/test_mode3_gpu.py -m gpu -q -x > gpurun_out/c20/tests.log 2>&1; rc=$?
tail -3 gpurun_out/c20/tests.log
[ $rc -ne 0 ] && exit $rc
ROUNDS=2 bash tools/ab_libs.sh c20 pylc_amd/libpylc_hip_prev.so pylc_amd/libpylc_hip.so 2>&1 | tee gpurun_out/c20/ab.txt
BENCH_ARGS="--config c5" ROUNDS=1 bash tools/ab_libs.sh c20c5 pylc_amd/libpylc_hip_prev.so pylc_amd/libpylc_hip.so 2>&1 | tee gpurun_out/c20/ab_c5.txt
BENCH_ARGS="--config c2" ROUNDS=1 bash tools/ab_libs.sh c20c2 pylc_amd/libpylc_hip_prev.so pylc_amd/libpylc_hip.so 2>&1 | tee gpurun_out/c20/ab_c2.txt
