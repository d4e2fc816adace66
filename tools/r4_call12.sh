#!/bin/bash
set -o pipefail
out=gpurun_out/r4m; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_fullsize_ops_gpu.py tests/test_nets_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do
  for v in "PYLC_WG_FLAGS=4 PYLC_WG_MAX_STEPS=0" "PYLC_WG_FLAGS=0 PYLC_WG_MAX_STEPS=256"; do
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $out/wg_ab.txt
  done
done
for c in c2 c5; do
for v in "PYLC_WG_FLAGS=4 PYLC_WG_MAX_STEPS=0" "PYLC_WG_FLAGS=0 PYLC_WG_MAX_STEPS=256"; do
  env $v timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-dp-overhead --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $v', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $out/wg_ab.txt
done
done
mkdir -p $out/pmc
bash tools/pmc_bench.sh $out/pmc 400 > $out/pmc.log 2>&1; tail -2 $out/pmc.log | head -1
python - <<'PY'
import json
t=json.load(open('gpurun_out/r4m/pmc/traffic.json'))
steps=7.0
tot=0
for k,v in t.items():
    if k.startswith('_'): continue
    if 'wgrad' in k or 'splitk' in k:
        gb=v['hbm_bytes_per_launch']*v['launches']/steps/1e9
        tot+=gb
        print('%-60s %5d launches/step %8.1f MB fetch %7.1f MB write  %6.2f GB/step' % (k[:60], v['launches']/steps, v['fetch_bytes_per_launch']/1e6, v['write_bytes_per_launch']/1e6, gb))
print('wgrad + split-K reduce: %.1f GB/step (round 3: 54 GB)' % tot)
PY
