export TMPDIR=/tmp
OUT=gpurun_out/r4q; mkdir -p $OUT
timeout 900 python bench.py --workload c4 --batch 8 --steps 1 --warmup 1 --no-preflight > $OUT/c4_n1.json 2> $OUT/c4_n1.err; echo "c4 N=1 rc=$? $(cut -c1-700 $OUT/c4_n1.json)"; grep -o '"group_results_rechecked_against_lone_calls": [^,}]*' $OUT/c4_n1.json; tail -2 $OUT/c4_n1.err | cut -c1-200
