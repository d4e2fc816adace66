#!/bin/bash
# bench.py with the one-accumulator <= 128-register 128x128 wgrad (PYLC_WGRAD_ACC1=1) and without (default), interleaved on one box
mkdir -p gpurun_out/r02_acc1
for v in base acc1 base2 acc1b; do
  if [ ${v:0:3} = acc ]; then export PYLC_WGRAD_ACC1=1; else unset PYLC_WGRAD_ACC1; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead "$@" > gpurun_out/r02_acc1/bench_$v.json 2> gpurun_out/r02_acc1/bench_$v.err || { tail -5 gpurun_out/r02_acc1/bench_$v.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r02_acc1/bench_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value'],1), 'tiles/s', round(d['ms_per_step'],2), 'ms frac', round(d['roofline']['frac'],3), {k: round(v['tflops']) for k, v in d['roofline']['by_kind'].items()}, d['config']['last_loss'])
"
done
