export TMPDIR=/tmp
OUT=gpurun_out/r4t; mkdir -p $OUT
MIXED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python tools/nar_batch_bench.py 32 > $OUT/batch32.log 2>&1
grep "U=" $OUT/batch32.log
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_batch32.csv
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
head -14 $OUT/kernel_stats_batch32.csv | cut -c1-200
