#!/bin/bash
set -o pipefail
out=gpurun_out/r4c; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_planes_gpu.py -q -k "specialised" > $out/ps_tests.log 2>&1; rc=$?
tail -5 $out/ps_tests.log
[ $rc -ne 0 ] && { echo "ps tests failed rc=$rc"; exit 1; }
timeout -k 10 500 python tools/ps_ab.py 20 2>&1 | grep -v amdgpu.ids | tee $out/ps_ab.txt
for i in 1 2; do
  for v in "PYLC_PS=0" "PYLC_PS=1" "PYLC_PS=3"; do
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $out/knob_ab.txt
  done
done
PYLC_PS=3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-dp-overhead 2>/dev/null | tail -1 > $out/bench_ps3.json
python -c "
import json; d=json.loads(open('$out/bench_ps3.json').read()); r=d['roofline']
print('ps3 bench', d['value'], d['ms_per_step'], r['frac'], {k: round(v['tflops']) for k, v in r['by_kind'].items()})"
