#!/bin/bash
# bench.py with the 1x1 wgrads held back to the next conv backward (PYLC_DEFER_WGRAD=1) vs launched right after their dgrad, interleaved on one box
mkdir -p gpurun_out/r02_defer
for v in base defer base2 defer2; do
  if [ ${v:0:2} = de ]; then export PYLC_DEFER_WGRAD=1; else unset PYLC_DEFER_WGRAD; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead > gpurun_out/r02_defer/bench_$v.json 2> gpurun_out/r02_defer/bench_$v.err || { tail -5 gpurun_out/r02_defer/bench_$v.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r02_defer/bench_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value'],1), 'tiles/s', round(d['ms_per_step'],2), 'ms frac', round(d['roofline']['frac'],3), {k: round(v['tflops']) for k, v in d['roofline']['by_kind'].items()}, d['config']['last_loss'])
"
done
