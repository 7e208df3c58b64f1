# absorb_kernel with LDS-staged 16-byte stores: parity tests + kernel stats of the single-request step (compare profiles/r5v)
exec < /dev/null      # nothing in a gpurun script may wait on stdin (a `head` without a file name once held a box until the call limit)
TAG=r5w; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity16.py -m gpu -x -q -k "absorb or xattn or cross or nar" 2>&1 | tail -5 > gpurun_out/$TAG/tests.txt
cat gpurun_out/$TAG/tests.txt
timeout 300 python tools/nar_step_bench.py "M5_NAR_DUAL=1" 2>&1 | tail -4 | tee gpurun_out/$TAG/step.txt
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o step -- python $R/tools/nar_step_bench.py "M5_NAR_DUAL=1" > /dev/null 2>&1
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv && head -9 "$f" < /dev/null | cut -c1-160
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
