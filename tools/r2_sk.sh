#!/bin/bash
# GEMM configuration check: default picks on all bench shapes + in-step timing.  usage: tools/r2_sk.sh <tag>
OUT=gpurun_out/${1:-r2sk}; mkdir -p $OUT
export M5_HIP_TOOLS=1
timeout 300 python tools/gemm_bench.py > $OUT/gemm_default.log 2>&1
grep -v "^$\|amdgpu.ids" $OUT/gemm_default.log
timeout 900 python tools/nar_step_bench.py ${NARAB:-"M5_GEMM_CFG_E4=6" "M5_NAR_ABSORB=1"} > $OUT/nar_ab.log 2>&1; echo "narab rc=$?"; grep round $OUT/nar_ab.log
