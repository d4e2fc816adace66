#!/bin/bash
# usage: tools/pmc.sh <outdir> <python args...>   -- four separate rocprofv3 --pmc passes (no trace domains)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/d -- python3 "$@" > /dev/null 2>&1
find $out -name "*counter_collection.csv" | head
