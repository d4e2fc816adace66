#!/bin/bash
# rocprofv3 kernel trace of tools/bn_table.py with wgrad on the main stream: serial duration of every BatchNorm kernel per grid size.  usage: bn_serial_trace.sh <tag> [c3|c5]
out=gpurun_out/$1; cfg=${2:-c3}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PYLC_SERIAL=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/bn_table.py $cfg > $out/bn_table_$cfg.txt 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
find $out/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/kernel_by_grid.py {} bn_ column_sum > $out/bn_by_grid_$cfg.txt 2>&1
rm -rf $out/trace
cat $out/bn_by_grid_$cfg.txt
