#!/bin/bash
# Matrix-pipe / LDS / occupancy counters of the bench's kernels: two rocprofv3 --pmc passes with the 8 SQ slots
# (+ GRBM_GUI_ACTIVE), kernel-trace / stats domains only, the program directly after `--`.   usage: pmc_mfma.sh <outdir> [bench args]
# Exits non-zero -- and keeps the raw CSVs -- when a pass fails, times out or leaves no counter_collection.csv: a partial pmc_mfma.json must
# not be attached to a bench line (bench.pmc_mfma only checks the build fingerprint).
set -o pipefail
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $out/counters_available.txt 2>&1
timeout 500 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d $out/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-dp-overhead "$@" > $out/bench_under_pmc.json 2> $out/pmc.err
rc1=$?; echo "sq pass rc=$rc1"
[ $rc1 -eq 0 ] && ls $out/sq/*/*counter_collection.csv > /dev/null 2>&1 || { echo "pmc_mfma: first pass failed (rc $rc1) or left no counter_collection.csv; raw output kept in $out"; exit 1; }
timeout 500 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES \
    --output-format csv -d $out/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-dp-overhead "$@" > /dev/null 2>> $out/pmc.err
rc2=$?; echo "sq2 pass rc=$rc2"
[ $rc2 -eq 0 ] && ls $out/sq2/*/*counter_collection.csv > /dev/null 2>&1 || { echo "pmc_mfma: second pass failed (rc $rc2) or left no counter_collection.csv; raw output kept in $out"; exit 1; }
python3 tools/pmc_mfma.py $out > $out/pmc_mfma.json || { echo "pmc_mfma.py failed; raw output kept in $out"; rm -f $out/pmc_mfma.json; exit 1; }
head -c 3000 $out/pmc_mfma.json
rm -rf $out/sq $out/sq2
