#!/bin/bash
# the data-parallel code path at world size 1 in four cells -- {one queue, wgrad side stream} x {torch.distributed, PYLC_COMM=native}
# -- each cell one bench.py run (the line's config.dp_codepath_overhead = the step with a one-rank RCCL group switched on against the same
# process's group-less step).  Usage: bash tools/dp_cells.sh [rounds]   -> gpurun_out/dp_cells/*.json and a table on stdout
set -o pipefail
rounds=${1:-1}
mkdir -p gpurun_out/dp_cells
for i in $(seq 1 $rounds); do
  for cell in "one_queue:torch:PYLC_COMM=torch" "one_queue:native:PYLC_COMM=native" "side_stream:torch:PYLC_SIDE_STREAM=1 PYLC_COMM=torch" "side_stream:native:PYLC_SIDE_STREAM=1 PYLC_COMM=native"; do
    name=${cell%%:*}; rest=${cell#*:}; comm=${rest%%:*}; setting=${rest#*:}
    f=gpurun_out/dp_cells/${name}_${comm}_$i
    env $setting $DP_CELLS_ENV timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 > $f.json 2> $f.err || { tail -5 $f.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$f.json").read().strip().splitlines()[-1])
c = d["config"]
print("round $i [%-12s %-7s] %.1f tiles/s %.2f ms | dp_codepath_overhead %s  collectives %s  wgrad_queue %s" % ("$name", "$comm", d["value"], d["ms_per_step"],
      ("%+.4f" % c["dp_codepath_overhead"]) if isinstance(c["dp_codepath_overhead"], float) else c["dp_codepath_overhead"], c["collectives_per_step"], c["wgrad_queue"]))
PY
  done
done
