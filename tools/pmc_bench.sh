#!/bin/bash
# HBM traffic of the bench's kernels: two separate, time-bounded rocprofv3 --pmc passes (FETCH_SIZE needs 3 TCC slots,
# WRITE_SIZE 2).  One HIP queue only (PYLC_NO_SIDE_STREAM): counter collection serialises kernels anyway.
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYLC_NO_SIDE_STREAM=1
timeout ${2:-400} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
echo "fetch pass rc=$?"
timeout ${2:-400} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
echo "write pass rc=$?"
python3 tools/pmc_traffic.py $out > $out/traffic.json
head -c 1200 $out/traffic.json
