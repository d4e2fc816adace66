#!/bin/bash
# Same-box comparison of several environment settings (the A/B knobs of pylc_amd/runtime.py and lib.py): bench.py (no CPU baseline, no DP
# leg) once per setting per round, interleaved.
# Usage: [BENCH_ARGS="--config c5"] bash tools/ab_multi.sh <rounds> "NAME=VAL ..." "NAME=VAL ..." ...     ("" = the defaults)
# e.g.   bash tools/ab_multi.sh 2 "" "PYLC_RUNTIME=no_relu_bits=1"          BENCH_ARGS="--config c5" bash tools/ab_multi.sh 2 "PYLC_RUNTIME=half_acts=0" ""
set -o pipefail
rounds=$1; shift
mkdir -p gpurun_out/ab_multi
for i in $(seq 1 $rounds); do
  k=0
  for setting in "$@"; do
    k=$((k+1))
    env $setting timeout -k 10 200 python bench.py --no-cpu-baseline --no-dp-overhead $BENCH_ARGS > gpurun_out/ab_multi/s${k}_$i.json 2> gpurun_out/ab_multi/s${k}_$i.err || { tail -5 gpurun_out/ab_multi/s${k}_$i.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_multi/s${k}_$i.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("round $i [%-28s]" % "${setting:-defaults}", round(d["value"], 1), "tiles/s", round(d["ms_per_step"], 2), "ms | frac", round(r["frac"], 3), {k: round(v["tflops"]) for k, v in r["by_kind"].items()})
PY
  done
done
