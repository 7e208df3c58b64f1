#!/bin/bash
# HBM traffic per launch of the dominant kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only (no other trace domains).
TAG=${1:-traffic}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp REP=2
export ONLY="${ONLY:-nar self qkv,nar out_proj,nar cross q,nar swiglu,nar linear2}"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/gemm_$c -o p -- python tools/gemm_bench.py > $OUT/gemm_$c.log 2>&1; echo "gemm $c rc=$?"
  [ "$NOATTN" = "1" ] || timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/attn_$c -o p -- python tools/attn_bench.py > $OUT/attn_$c.log 2>&1; echo "attn $c rc=$?"
done
python tools/pmc_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.csv" -size +3M -delete
