export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r4i}
mkdir -p $OUT
for nb in 8 16; do
  timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight --nar-batch $nb > $OUT/c3_nb$nb.json 2> $OUT/c3_nb$nb.err; echo "c3 nar_batch=$nb rc=$? $(cut -c1-120 $OUT/c3_nb$nb.json)"; grep -o '"time_split_s_per_step": {[^}]*}' $OUT/c3_nb$nb.json;  grep -o '"last_nar_batch": {[^}]*}' $OUT/c3_nb$nb.json; tail -2 $OUT/c3_nb$nb.err | cut -c1-200
done
M5_NAR_ROWTILES=0 timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight --nar-batch 8 > $OUT/c3_nb8_nort.json 2> $OUT/c3_nort.err; echo "c3 nar_batch=8 no row tiles rc=$? $(cut -c1-120 $OUT/c3_nb8_nort.json)"; grep -o '"time_split_s_per_step": {[^}]*}' $OUT/c3_nb8_nort.json
M5_NAR_DLN=0 timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight --nar-batch 8 > $OUT/c3_nb8_nodln.json 2> $OUT/c3_nodln.err; echo "c3 nar_batch=8 no dln rc=$? $(cut -c1-120 $OUT/c3_nb8_nodln.json)"; grep -o '"time_split_s_per_step": {[^}]*}' $OUT/c3_nb8_nodln.json
