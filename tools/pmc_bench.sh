#!/bin/bash
# HBM traffic of the bench's kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python3 tools/pmc_traffic.py $out > $out/traffic.json
cat $out/traffic.json | head -c 1500
