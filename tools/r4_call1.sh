#!/bin/bash
# round-4 first GPU call: whole -m gpu suite, clean bench line, wgrad launch-order A/B, in-step conv table, c5 inference line
set -o pipefail
out=gpurun_out/r4a; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/gputests.log 2>&1; rc=$?
echo "pytest exit $rc" >> $out/gputests.log; tail -5 $out/gputests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > $out/bench_n1.json 2> $out/bench.err || exit $?
python - <<'PY'
import json; d=json.load(open('gpurun_out/r4a/bench_n1.json'))
print('bench', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], 'instr ms', d['roofline']['instrumented_ms_per_step'], 'hbm', d['roofline']['hbm']['achieved'], d['roofline']['hbm']['ms_per_step'])
PY
for i in 1 2; do
  for h in 0 1; do
    PYLC_WGRAD_HOLD=$h timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hold=$h', d['value'], d['ms_per_step'])" | tee -a $out/hold_ab.txt
  done
done
timeout -k 10 200 python tools/conv_table.py 2>/dev/null | grep -v amdgpu.ids > $out/conv_table_in_step.txt; head -30 $out/conv_table_in_step.txt
timeout -k 10 300 python bench.py --config c5 --inference > $out/bench_c5_inference.json 2> $out/bench_c5_inference.err || exit $?
head -c 600 $out/bench_c5_inference.json; echo
