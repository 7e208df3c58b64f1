export TMPDIR=/tmp REP=4
echo "== residual, producer form (cfg 0 vs 7)"; DLN=1 SWEEP=0,7 ONLY="big out_proj,big linear2,huge out_proj,huge linear2" timeout 400 python tools/gemm_bench.py 2>&1 | grep -E "out_proj|linear2"
echo "== swiglu / qkv (cfg 0 1 2 5)"; SWEEP=0,1,2,5 ONLY="big swiglu,big qkv,huge swiglu,huge qkv" timeout 600 python tools/gemm_bench.py 2>&1 | grep -E "swiglu|qkv"
