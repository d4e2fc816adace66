#!/bin/bash
# One GPU-box call: conv / net parity tests, the per-tile phase stamps, the per-shape conv benchmark and the full step.
set -o pipefail
mkdir -p gpurun_out
tag=${1:-chk}
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "pytest exit $rc" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi      # killed / timed out: no further GPU step
for sh in "256 1024 1 0 32 32" "64 256 1 0 32 128" "256 256 3 1 32 32" "256 256 3 1 32 128"; do
    echo "== $sh"; PP_FLAGS=1024 PYLC_PLANES=1 timeout -k 10 120 python tools/pp_stamps.py $sh 2>&1 | grep "tile phases" | cut -c1-300 || exit 1
done > gpurun_out/${tag}_stamps.txt 2>&1
cat gpurun_out/${tag}_stamps.txt
PYLC_PLANES=1 timeout -k 10 150 python tools/conv_bench.py fwd dgrad > gpurun_out/${tag}_convbench.txt 2>&1 || exit $?
cat gpurun_out/${tag}_convbench.txt
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(round(d['value'], 1), 'tiles/s', round(r['frac'], 4), round(r['avg_launch_ms'], 4), {k: round(v['tflops']) for k, v in r['by_kind'].items()})
PY
