#!/bin/bash
set -o pipefail
out=gpurun_out/r4k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fl in 0 1 2; do
  timeout -k 10 200 python3 tools/wgrad_traffic.py $fl 2>/dev/null | grep flags | tee -a $out/wgrad_times.txt
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc$fl -- python3 tools/wgrad_traffic.py $fl > /dev/null 2>&1
  echo "== flags $fl"; python3 tools/wgrad_traffic.py --parse $out/pmc$fl | tee -a $out/wgrad_fetch_$fl.txt
  rm -rf $out/pmc$fl
done
