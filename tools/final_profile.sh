#!/bin/bash
# Round-end evidence on one GPU box: the whole -m gpu suite, the other configurations, bench.py (with the CPU baseline),
# the rocprofv3 kernel trace of the same bench command, and the two PMC passes for HBM traffic.
set -o pipefail
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
    timeout -k 10 700 python -m pytest tests -m gpu -x -q > $out/gputests.log 2>&1
    rc=$?
    echo "pytest exit $rc" >> $out/gputests.log
    tail -4 $out/gputests.log
    if [ $rc -ne 0 ]; then exit $rc; fi
fi
timeout -k 10 300 python tools/net_bench.py 2> $out/other_configs.err | grep -v amdgpu.ids > $out/other_configs.txt || exit $?
cat $out/other_configs.txt
timeout -k 10 300 python bench.py > $out/bench_n1.json 2> $out/bench.err || exit $?
tail -c 400 $out/bench_n1.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err || exit $?
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/trace_streams.py {} > $out/trace_streams.txt 2>&1
rm -rf $out/trace
head -12 $out/kernel_stats.csv | cut -c1-160
bash tools/pmc_bench.sh $out/pmc 400
rm -rf $out/pmc/fetch $out/pmc/write
