#!/bin/bash
set -o pipefail
out=gpurun_out/r4k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4 0" "8 256" "0 256"; do
  set -- $cfg; r=$1; steps=$2
  fl=$((steps * 16 + r))
  timeout -k 10 200 python3 tools/wgrad_traffic.py $fl 2>/dev/null | grep flags | sed "s/flags $fl/raster $r max_steps $steps/" | tee -a $out/wgrad_final_times.txt
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc$fl -- python3 tools/wgrad_traffic.py $fl > /dev/null 2>&1
  echo "== raster $r max_steps $steps"; python3 tools/wgrad_traffic.py --parse $out/pmc$fl | tee -a $out/wgrad_final_fetch_${r}_$steps.txt
  rm -rf $out/pmc$fl
done
