# round 6, call a: new kernels (fp32 split-f16 GEMM, known-row fast path of nar_sample), probes (8-wave residual region, attention setprio)
exec < /dev/null      # nothing in a gpurun script may wait on stdin
TAG=r6a; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "f32_split or nar_sample or gemm_epilogues or nar_uniforms" 2>&1 | tail -25 > gpurun_out/$TAG/tests_kernels.txt
tail -3 gpurun_out/$TAG/tests_kernels.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "nar_tiny_logits or nar_tiny_f32" 2>&1 | tail -30 > gpurun_out/$TAG/tests_e2e.txt
tail -3 gpurun_out/$TAG/tests_e2e.txt
timeout 900 python tools/f32_products_bench.py > gpurun_out/$TAG/f32_products.txt 2>&1
tail -14 gpurun_out/$TAG/f32_products.txt
DLN=1 ONLY="nar out_proj,nar p.b,nar linear2,big out_proj,big linear2" SWEEP=7,11 timeout 400 python tools/gemm_bench.py > gpurun_out/$TAG/gemm_cfg11.txt 2>&1
cat gpurun_out/$TAG/gemm_cfg11.txt | cut -c1-150
timeout 500 python tools/nar_step_bench.py "M5_GEMM_CFG_E2=7" "M5_GEMM_CFG_E2=11" 2>&1 | tail -4 > gpurun_out/$TAG/nar_step_cfg11.txt
cat gpurun_out/$TAG/nar_step_cfg11.txt
for v in 0 4 0 4; do echo "== M5_ATTN_VARIANT=$v"; M5_ATTN_VARIANT=$v timeout 200 python tools/attn_bench.py 2>&1 | head -2; done > gpurun_out/$TAG/attn_setprio.txt
cat gpurun_out/$TAG/attn_setprio.txt
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o step -- python $R/tools/nar_step_bench.py "M5_NAR_DUAL=1" > /dev/null 2>&1
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv && head -16 "$f" < /dev/null | cut -c1-150
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
