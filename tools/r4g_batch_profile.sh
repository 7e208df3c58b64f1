export TMPDIR=/tmp
OUT=gpurun_out/r4g
for d in 1 0; do
  M5_NAR_DLN=$d MIXED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dln$d -o p -- python tools/nar_batch_bench.py 8 16 > $OUT/batch_dln$d.log 2>&1
  grep "U=" $OUT/batch_dln$d.log
  find $OUT/prof_dln$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_batch_dln$d.csv
  find $OUT/prof_dln$d -name "*kernel_trace.csv" -delete; find $OUT/prof_dln$d -name "*.db" -delete
done
head -25 $OUT/kernel_stats_batch_dln1.csv | cut -c1-220
