export TMPDIR=/tmp
OUT=gpurun_out/r4j
mkdir -p $OUT
timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight --nar-batch 32 --nar-in-flight 1 > $OUT/c3_nb32.json 2> $OUT/c3_nb32.err; echo "c3 nar_batch=32 rc=$? $(cut -c1-120 $OUT/c3_nb32.json)"; grep -o '"time_split_s_per_step": {[^}]*}' $OUT/c3_nb32.json
