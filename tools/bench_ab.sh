#!/bin/bash
# bench.py with fp16-plane activations (default) and with PYLC_NO_PLANES=1, interleaved on one box; prints one summary line per run.
set -o pipefail
tag=${1:-ab}; shift
mkdir -p gpurun_out/$tag
for v in ${@:-planes noplanes planes2 noplanes2}; do
  if [ ${v:0:2} = no ]; then export PYLC_NO_PLANES=1; else unset PYLC_NO_PLANES; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/bench_$v.err || { tail -5 gpurun_out/$tag/bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/$tag/bench_$v.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$v", round(d["value"], 1), "tiles/s", round(d["ms_per_step"], 2), "ms | frac", round(r["frac"], 3), {k: round(v["tflops"]) for k, v in r["by_kind"].items()},
      "conversions/step", d["config"].get("planes_to_fp32_conversions_per_step"), "amax passes", d["config"]["standalone_range_passes_per_step"], "loss", [round(x, 4) for x in d["config"]["last_loss"]])
PY
done
