# round 6, call b: fp32 split-f16 attention + 64-row GEMM tiles, RCCL on one GPU, store-first producer epilogue A/B, heads GEMM tilings
exec < /dev/null
TAG=r6b; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "f32_split or attention or nar_sample or gemm_epilogues" 2>&1 | tail -40 > gpurun_out/$TAG/tests_kernels.txt
tail -4 gpurun_out/$TAG/tests_kernels.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "nar_tiny_logits or nar_tiny_f32 or full_size_goldens_f32" 2>&1 | tail -30 > gpurun_out/$TAG/tests_e2e.txt
tail -4 gpurun_out/$TAG/tests_e2e.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "rccl" 2>&1 | tail -30 > gpurun_out/$TAG/tests_rccl.txt
tail -6 gpurun_out/$TAG/tests_rccl.txt
timeout 900 python tools/f32_products_bench.py > gpurun_out/$TAG/f32_products.txt 2>&1
tail -28 gpurun_out/$TAG/f32_products.txt
for rep in 1 2; do for lib in libmars5_hip_tools.so libmars5_hip_tools_sf.so; do
  echo "== $lib" >> gpurun_out/$TAG/store_first_ab.txt
  M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/$lib timeout 400 python tools/nar_step_bench.py "M5_NAR_DUAL=1" 2>&1 | tail -2 >> gpurun_out/$TAG/store_first_ab.txt
done; done
cat gpurun_out/$TAG/store_first_ab.txt
ONLY="nar heads folded" SWEEP=0,1,2,5 timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/heads_sweep.txt
cut -c1-150 gpurun_out/$TAG/heads_sweep.txt
R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o step -- python $R/tools/f32_products_bench.py > /dev/null 2>&1
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/f32_kernel_stats.csv && head -14 "$f" < /dev/null | cut -c1-170
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
