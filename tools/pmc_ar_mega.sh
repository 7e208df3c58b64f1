#!/bin/bash
# HBM-side traffic of one persistent decode step (ar_mega_kernel): FETCH_SIZE and WRITE_SIZE in separate passes,
# kernel-trace only, over the 12 eager steps of tools/ar_mega_clock.py.  usage: tools/pmc_ar_mega.sh <tag>
TAG=${1:-pmc_mega}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python tools/ar_mega_clock.py > $OUT/$c.log 2>&1; echo "$c rc=$?"
done
python tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A3 "ar_mega\|gemv_stream\|sample_kernel" $OUT/summary.txt | head -40
find $OUT -name "*.csv" -size +3M -delete
