#!/bin/bash
# rocprofv3 kernel trace of bench.py: per-queue busy time per kernel (tools/trace_streams.py) + the stats CSV.   usage: trace_bench.sh <tag> [bench args]
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/rocprof.err
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/trace_streams.py {} 24 > $out/trace_streams.txt 2>&1
rm -rf $out/trace
cat $out/trace_streams.txt
