#!/bin/bash
set -o pipefail
out=gpurun_out/r4j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_nets_gpu.py -q -k "distributed_code_path or ranks_equal" > $out/dist_tests.log 2>&1; rc=$?
tail -8 $out/dist_tests.log
[ $rc -ne 0 ] && exit 1
for v in "PYLC_COMM=" "PYLC_COMM=native"; do
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2), 'dp overhead', d['config']['dp_codepath_overhead'], d['config']['collectives_per_step'])" | tee -a $out/comm_ab.txt
done
