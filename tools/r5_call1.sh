#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c1
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_mode3_gpu.py tests/test_round3_gpu.py -m gpu -x -q > gpurun_out/c1/tests.log 2>&1; rc=$?
tail -5 gpurun_out/c1/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_libs.sh c1 pylc_amd/libpylc_hip_r4.so pylc_amd/libpylc_hip.so pylc_amd/libpylc_hip_pk.so 2>&1 | tee gpurun_out/c1/ab.txt
