# round 6, call w: operand build + uniform draws as one heterogeneous launch (m5_xattn_absorb_uniforms) against the two launches
exec < /dev/null
TAG=r6w; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 500 python tools/nar_step_bench.py "M5_NAR_HEADFUSE=0" "M5_NAR_HEADFUSE=1" 2>&1 | tail -4 > gpurun_out/$TAG/nar_step_headfuse_ab.txt
cat gpurun_out/$TAG/nar_step_headfuse_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py tests/test_gpu_parity16.py -m gpu -q -k "uniform or c_composed or absorbed or nar_sample" 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/$TAG/tests.txt
cat gpurun_out/$TAG/tests.txt
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o step -- python $R/tools/nar_step_bench.py "M5_NAR_HEADFUSE=1" > $R/gpurun_out/$TAG/step_prof_run.txt 2>&1
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/nar_step_kernel_stats.csv && grep -E "absorb|uniform|nar_sample" "$f" | cut -c1-200
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
