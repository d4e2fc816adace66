#!/bin/bash
# round-4 second GPU call: new-kernel bit-identity first (short, bounded), then its A/B, then the whole suite and the bench A/Bs
set -o pipefail
out=gpurun_out/r4b; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_planes_gpu.py -q -k "specialised" > $out/ps_tests.log 2>&1; rc=$?
tail -15 $out/ps_tests.log
[ $rc -ne 0 ] && { echo "ps tests failed rc=$rc"; PS_BAD=1; }
if [ -z "$PS_BAD" ]; then
  timeout -k 10 400 python tools/ps_ab.py 20 2>&1 | grep -v amdgpu.ids | tee $out/ps_ab.txt
fi
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_planes_gpu.py::test_specialised_wave_1x1_kernel_is_bit_identical --deselect tests/test_planes_gpu.py::test_specialised_wave_1x1_kernel_in_a_bottleneck_chain > $out/gputests.log 2>&1
echo "pytest exit $?" >> $out/gputests.log; tail -12 $out/gputests.log
grep -E "^(FAILED|ERROR)" $out/gputests.log | head -30
timeout -k 10 300 python bench.py > $out/bench_n1.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<'PY'
import json; d=json.load(open('gpurun_out/r4b/bench_n1.json'))
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'roofline', r['frac'], 'instr ms', r['instrumented_ms_per_step'], 'hbm', r['hbm']['achieved'], r['hbm']['ms_per_step'])
print({k: round(v['tflops']) for k, v in r['by_kind'].items()})
PY
for i in 1 2; do
  for v in "PYLC_WGRAD_HOLD=0 PYLC_PS=0" "PYLC_WGRAD_HOLD=1 PYLC_PS=0" "PYLC_WGRAD_HOLD=1 PYLC_PS=1" "PYLC_WGRAD_HOLD=1 PYLC_PS=3"; do
    [ -n "$PS_BAD" ] && [[ "$v" != *"PYLC_PS=0" ]] && continue
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $out/knob_ab.txt
  done
done
timeout -k 10 200 python tools/conv_table.py 2>/dev/null | grep -v amdgpu.ids > $out/conv_table_in_step.txt; head -24 $out/conv_table_in_step.txt
timeout -k 10 300 python bench.py --config c5 --inference > $out/bench_c5_inference.json 2> $out/bench_c5_inference.err || exit $?
head -c 400 $out/bench_c5_inference.json; echo
