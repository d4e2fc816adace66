#!/bin/bash
# bench.py at several BatchNorm row-slab counts (PYLC_MAX_SLABS), interleaved on one box: 768 (the default) re-confirmed in round 2
mkdir -p gpurun_out/r02_slabs
for v in 768 512 640 896 1024 768b; do
  export PYLC_MAX_SLABS=${v%b}
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead > gpurun_out/r02_slabs/bench_$v.json 2> gpurun_out/r02_slabs/bench_$v.err || { tail -5 gpurun_out/r02_slabs/bench_$v.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r02_slabs/bench_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value'],1), 'tiles/s', round(d['ms_per_step'],2), 'ms')
"
done
