set -o pipefail
o=gpurun_out/${1:-r06_final_b}; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1 || { tail -5 $o/smoke.txt; exit 1; }
tail -2 $o/smoke.txt
timeout -k 10 300 python bench.py > $o/bench_n1.json 2> $o/bench.err || exit 1
for c in c2 c4 c5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-dp-overhead > $o/bench_$c.json 2> $o/bench_$c.err || exit 1; done
PYLC_SERIAL=1 timeout -k 10 300 python tools/conv_table.py 2>/dev/null | grep -v amdgpu.ids > $o/conv_table_serial.txt || exit 1
timeout -k 10 300 python tools/conv_table.py 2>/dev/null | grep -v amdgpu.ids > $o/conv_table_in_step.txt || exit 1
PYLC_SERIAL=1 PYLC_TABLE_CFG=c2 timeout -k 10 300 python tools/conv_table.py 2>/dev/null | grep -v amdgpu.ids > $o/conv_table_c2_serial.txt || exit 1
bash tools/trace_bench.sh ${1:-r06_final_b}/c2 --config c2 --no-dp-overhead > /dev/null 2>&1 || exit 1
bash tools/trace_bench.sh ${1:-r06_final_b}/c5 --config c5 --no-dp-overhead > /dev/null 2>&1 || exit 1
head -3 $o/c2/trace_streams.txt
timeout -k 10 300 python tools/bn_table.py 2>/dev/null | grep -v amdgpu.ids > $o/bn_table_in_step.txt || exit 1
PYLC_SERIAL=1 timeout -k 10 300 python tools/bn_table.py 2>/dev/null | grep -v amdgpu.ids > $o/bn_table_serial.txt || exit 1
timeout -k 10 400 python tools/eval_ab.py 2>/dev/null | grep -v amdgpu.ids > $o/eval_ab.txt || exit 1
cat $o/eval_ab.txt | cut -c1-200
# the driver's own N = 2 command, two ranks sharing this box's GPU over gloo (RCCL refuses two ranks on one device): the data-parallel bench path end to end
PYLC_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 2> $o/bench_n2_gloo.err | tail -1 > $o/bench_n2_gloo_rehearsal.json || exit 1
head -c 400 $o/bench_n2_gloo_rehearsal.json; echo
