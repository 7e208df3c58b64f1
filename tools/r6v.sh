# round 6, call v: product-build GEMM tests with loader waves on, step / token times, kernel stats of the lone and the batched NAR step
exec < /dev/null
TAG=r6v; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity16.py -m gpu -q -k "gemm or deferred or row_tile or nar_full_size or c_composed" 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/$TAG/tests_gemm.txt
cat gpurun_out/$TAG/tests_gemm.txt
timeout 300 python tools/ar_step_bench.py "X=1" 2>&1 | grep round > gpurun_out/$TAG/ar_step.txt; cat gpurun_out/$TAG/ar_step.txt
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o step -- python $R/tools/nar_step_bench.py "X=1" > $R/gpurun_out/$TAG/step_prof_run.txt 2>&1
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/nar_step_kernel_stats.csv && head -22 "$f" | cut -c1-170
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
cd /tmp
GRAPH=0 MIXED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/profb -o batch -- python $R/tools/nar_batch_bench.py 16 > $R/gpurun_out/$TAG/batch_prof_run.txt 2>&1
cd $R; grep "U=" gpurun_out/$TAG/batch_prof_run.txt
f=$(find gpurun_out/$TAG/profb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/nar_batch16_kernel_stats.csv && head -22 "$f" | cut -c1-170
find gpurun_out/$TAG/profb -type f ! -name '*stats.csv' -delete
MIXED=1 timeout 300 python tools/nar_batch_bench.py 16 32 2>&1 | grep "U=" > gpurun_out/$TAG/batch_graph.txt; cat gpurun_out/$TAG/batch_graph.txt
