#!/bin/bash
set -o pipefail
out=gpurun_out/r4k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for steps in 0 512 256 128 64; do
  fl=$((steps * 4))
  timeout -k 10 200 python3 tools/wgrad_traffic.py $fl 2>/dev/null | grep flags | sed "s/flags $fl/max_steps $steps/" | tee -a $out/wgrad_cap_times.txt
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc$fl -- python3 tools/wgrad_traffic.py $fl > /dev/null 2>&1
  echo "== max_steps $steps"; python3 tools/wgrad_traffic.py --parse $out/pmc$fl | tee -a $out/wgrad_cap_fetch_$steps.txt
  rm -rf $out/pmc$fl
done
