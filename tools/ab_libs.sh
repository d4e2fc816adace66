#!/bin/bash
# Same-box A/B of several builds of the library: bench.py with PYLC_LIB=<each .so>, interleaved, ROUNDS times.
# Prints tiles/s, ms/step and the live per-kind TFLOP/s of the dominant conv family (roofline.by_kind) per run.
# usage: [BENCH_ARGS="--config c5"] [ROUNDS=2] bash tools/ab_libs.sh tag libA.so libB.so ...
set -o pipefail
tag=$1; shift
rounds=${ROUNDS:-2}
mkdir -p gpurun_out/$tag
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    name=$(basename $lib .so)_$r
    PYLC_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-dp-overhead $BENCH_ARGS > gpurun_out/$tag/$name.json 2>> gpurun_out/$tag/bench.err || exit $?
    python - <<PY
import json
d = json.loads(open('gpurun_out/$tag/$name.json').read().strip().splitlines()[-1])
r = d.get('roofline', {})
kinds = ' '.join('%s %.0f' % (k, v['tflops']) for k, v in sorted(r.get('by_kind', {}).items()))
print('%-24s %7.1f tiles/s %7.2f ms | family frac %.3f avg %.1f us | %s | bn %.1f ms' % ('$name', d['value'], d['ms_per_step'], r.get('frac', 0), 1e3 * r.get('avg_launch_ms', 0), kinds, r.get('hbm', {}).get('ms_per_step', 0)), flush=True)
PY
  done
done
