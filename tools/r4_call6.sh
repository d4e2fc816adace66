#!/bin/bash
set -o pipefail
out=gpurun_out/r4f; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_eval_planes_gpu.py -q -x > $out/eval_planes_tests.log 2>&1; rc=$?
tail -25 $out/eval_planes_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/eval_ab.py 2>&1 | grep -v amdgpu.ids | tee $out/eval_ab.txt
timeout -k 10 300 python bench.py --config c5 --inference 2>/dev/null | tail -1 > $out/bench_c5_inference.json; head -c 700 $out/bench_c5_inference.json; echo
