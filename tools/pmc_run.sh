#!/bin/bash
# usage: tools/pmc_run.sh <tag> <python script + args...>   -- SQ/TCC PMC passes (kernel-trace only)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp REP=2
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
CMD="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run sq3 SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_EXP_GDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM
run grbm GRBM_GUI_ACTIVE
python tools/pmc_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.csv" -size +3M -delete
