#!/bin/bash
set -o pipefail
out=gpurun_out/r4l; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_fullsize_ops_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do
  for v in "PYLC_WG_FLAGS=4 PYLC_WG_MAX_STEPS=0" "PYLC_WG_FLAGS=0 PYLC_WG_MAX_STEPS=256" "PYLC_WG_FLAGS=0 PYLC_WG_MAX_STEPS=128"; do
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $out/wg_ab.txt
  done
done
for v in "PYLC_WG_FLAGS=4 PYLC_WG_MAX_STEPS=0" "PYLC_WG_FLAGS=0 PYLC_WG_MAX_STEPS=256"; do
  env $v timeout -k 10 200 python bench.py --config c2 --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 $v', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $out/wg_ab.txt
done
