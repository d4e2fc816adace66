#!/bin/bash
# rocprofv3 kernel trace of bench.py: per-kernel totals (tools/trace_streams.py) and, per kernel name, launches, average duration and the idle
# gap behind each launch on its queue (tools/kernel_by_grid.py with an empty filter).   usage: trace_gaps.sh <tag> [bench args]
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dp-overhead "$@" > $out/bench_under_rocprof.json 2> $out/rocprof.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_streams.py $f 30 > $out/trace_streams.txt 2>&1
python3 tools/kernel_by_grid.py $f "" > $out/by_grid.txt 2>&1
rm -rf $out/trace
cat $out/trace_streams.txt
