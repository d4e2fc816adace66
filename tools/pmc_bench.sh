#!/bin/bash
# HBM traffic of the bench's kernels: two separate, time-bounded rocprofv3 --pmc passes (FETCH_SIZE needs 3 TCC slots,
# WRITE_SIZE 2).  One HIP queue only (PYLC_NO_SIDE_STREAM): counter collection serialises kernels anyway.
# Exits non-zero (raw CSVs kept) when a pass fails or leaves no counter_collection.csv.   usage: pmc_bench.sh <outdir> [timeout] [bench args]
set -o pipefail
out=$1; to=${2:-400}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYLC_NO_SIDE_STREAM=1
timeout $to rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-dp-overhead "$@" > /dev/null 2> $out.fetch.err
rc=$?; echo "fetch pass rc=$rc"
[ $rc -eq 0 ] && ls $out/fetch/*/*counter_collection.csv > /dev/null 2>&1 || { echo "pmc_bench: fetch pass failed (rc $rc) or left no counter_collection.csv"; exit 1; }
timeout $to rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-dp-overhead "$@" > /dev/null 2> $out.write.err
rc=$?; echo "write pass rc=$rc"
[ $rc -eq 0 ] && ls $out/write/*/*counter_collection.csv > /dev/null 2>&1 || { echo "pmc_bench: write pass failed (rc $rc) or left no counter_collection.csv"; exit 1; }
python3 tools/pmc_traffic.py $out > $out/traffic.json || { echo "pmc_traffic.py failed"; rm -f $out/traffic.json; exit 1; }
head -c 1200 $out/traffic.json
