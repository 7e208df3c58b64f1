#!/bin/bash
# PMC passes over the GEMM microbench (counters only with --kernel-trace, as gpurun requires).
TAG=${1:-pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export SWEEP=${SWEEP:-0} ONLY=${ONLY:-"nar out_proj,nar swiglu,enc swiglu (hoisted)"} REP=3
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python tools/gemm_bench.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUTDIR", "")
for d in sorted(glob.glob("%s/*/" % "gpurun_out/%s" % os.environ.get("TAGX","pmc"))):
    pass
PY
