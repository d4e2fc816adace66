#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c12
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c12/tests.log 2>&1; rc=$?
tail -4 gpurun_out/c12/tests.log
[ $rc -ne 0 ] && exit $rc
PYLC_LIB=$PWD/pylc_amd/libpylc_hip_exp.so timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_round3_gpu.py -m gpu -q -k "persistent or specialised or cu_masked or bn_backward_sums" > gpurun_out/c12/tests_exp.log 2>&1; rc=$?
tail -4 gpurun_out/c12/tests_exp.log
exit $rc
