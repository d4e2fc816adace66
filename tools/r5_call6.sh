#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c6
timeout -k 10 900 python -m pytest tests/test_round3_gpu.py tests/test_planes_gpu.py "tests/test_nets_gpu.py::test_distributed_code_path_single_rank" -m gpu -x -q > gpurun_out/c6/tests.log 2>&1; rc=$?
tail -5 gpurun_out/c6/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_multi.sh 2 "" "PYLC_FUSE_BN_SUMS=1" 2>&1 | tee gpurun_out/c6/ab_fuse.txt
