export TMPDIR=/tmp
OUT=gpurun_out/r4s; mkdir -p $OUT
timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$? $(cut -c1-160 $OUT/c3.json)"; grep -o '"time_split_s_per_step": {[^}]*}' $OUT/c3.json
timeout 600 python bench.py --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-batch-leg --no-preflight > $OUT/bench_f16.json 2> $OUT/f16.err; echo "f16 rc=$? $(tail -1 $OUT/bench_f16.json | cut -c1-200)"
timeout 400 python tools/blas_yardstick.py > $OUT/yard.log 2>&1; echo "yard rc=$?"; tail -14 $OUT/yard.log
