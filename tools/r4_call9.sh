#!/bin/bash
set -o pipefail
out=gpurun_out/r4i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py tests/test_mode3_gpu.py tests/test_eval_planes_gpu.py tests/test_round3_gpu.py -q > $out/mode3_tests.log 2>&1; rc=$?
tail -6 $out/mode3_tests.log; grep -E "^(FAILED|ERROR)" $out/mode3_tests.log | head
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/eval_ab.py xception_1024_gray_bs8_mode3 2>&1 | grep -v amdgpu.ids | tee $out/eval_ab.txt
timeout -k 10 300 python bench.py --config c5 --inference 2>/dev/null | tail -1 > $out/bench_c5_inference.json; python -c "
import json; d=json.loads(open('$out/bench_c5_inference.json').read()); print('c5 inference', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})"
timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline --no-dp-overhead 2>/dev/null | tail -1 > $out/bench_c5.json; python -c "
import json; d=json.loads(open('$out/bench_c5.json').read()); print('c5 train', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})"
