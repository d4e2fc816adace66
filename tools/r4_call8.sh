#!/bin/bash
set -o pipefail
out=gpurun_out/r4h; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1
echo "pytest exit $?" >> $out/gputests.log; tail -8 $out/gputests.log
grep -E "^(FAILED|ERROR)" $out/gputests.log | head -30
