#!/bin/bash
# Round-end evidence on one GPU box: the whole -m gpu suite, the other configurations, bench.py (with the CPU baseline),
# the rocprofv3 kernel trace of the same bench command, and the two PMC passes for HBM traffic.
set -o pipefail
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
    timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gputests.log 2>&1
    rc=$?
    echo "pytest exit $rc" >> $out/gputests.log
    tail -4 $out/gputests.log
    if [ $rc -ne 0 ]; then exit $rc; fi
fi
timeout -k 10 300 python tools/net_bench.py 2> $out/other_configs.err | grep -v amdgpu.ids > $out/other_configs.txt || exit $?
cat $out/other_configs.txt
timeout -k 10 300 python bench.py > $out/bench_n1.json 2> $out/bench.err || exit $?
tail -c 400 $out/bench_n1.json; echo
for c in c2 c4 c5; do
    timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-dp-overhead > $out/bench_$c.json 2> $out/bench_$c.err || exit $?
    head -c 300 $out/bench_$c.json; echo
done
timeout -k 10 300 python bench.py --config c5 --inference > $out/bench_c5_inference.json 2> $out/bench_c5_inference.err || exit $?
head -c 300 $out/bench_c5_inference.json; echo
timeout -k 10 200 python tools/pl_dephase_ab.py 20 2> /dev/null | grep -v amdgpu.ids > $out/conv_pl_shapes.txt
timeout -k 10 200 python tools/aten_ops_in_step.py 2> /dev/null | grep -v amdgpu.ids > $out/aten_ops_in_step.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dp-overhead > $out/bench_under_rocprof.json 2> $out/rocprof.err || exit $?
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/trace_streams.py {} > $out/trace_streams.txt 2>&1
rm -rf $out/trace
head -12 $out/kernel_stats.csv | cut -c1-160
mkdir -p $out/pmc
bash tools/pmc_bench.sh $out/pmc 400 || { echo "final_profile: HBM-traffic PMC passes failed (raw output kept under $out/pmc)"; exit 1; }
rm -rf $out/pmc/fetch $out/pmc/write
bash tools/pmc_mfma.sh $out/pmc_mfma > $out/pmc_mfma.log 2>&1 || { tail -5 $out/pmc_mfma.log; echo "final_profile: matrix-pipe PMC passes failed"; exit 1; }
tail -3 $out/pmc_mfma.log
