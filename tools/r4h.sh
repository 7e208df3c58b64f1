export TMPDIR=/tmp
OUT=gpurun_out/r4h
mkdir -p $OUT
for nb in 8 16; do
  timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight --nar-batch $nb > $OUT/c3_nb$nb.json 2> $OUT/c3_nb$nb.err; echo "c3 nar_batch=$nb rc=$? $(cut -c1-200 $OUT/c3_nb$nb.json)"; grep -o '"time_split_s_per_step": {[^}]*}' $OUT/c3_nb$nb.json
done
M5_NAR_ROWTILES=0 timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight --nar-batch 8 > $OUT/c3_nb8_nort.json 2> $OUT/c3_nort.err; echo "c3 nar_batch=8 no row tiles rc=$? $(cut -c1-200 $OUT/c3_nb8_nort.json)"
echo "== gemm default"; ONLY="nar self qkv,nar swiglu,big swiglu,big qkv" timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "qkv|swiglu"
echo "== gemm 32-deep K-steps (cfg 9 / 10)"; M5_GEMM_CFG_E3=9 M5_GEMM_CFG_E4=10 ONLY="nar self qkv,nar swiglu,big swiglu,big qkv" timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "qkv|swiglu"
