# round 6, call d: C-plan bit-identity test, nar_sample after the uniform prefetch, PMC traffic of the residual class (this round's build)
exec < /dev/null
TAG=r6d; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity16.py tests/test_gpu_kernels.py -m gpu -q -s -k "c_composed or nar_sample or failure" 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/$TAG/tests.txt
cat gpurun_out/$TAG/tests.txt | cut -c1-220
DLN=1 ONLY="nar out_proj,nar p.b,nar linear2" NOATTN=1 bash tools/pmc_traffic.sh $TAG/pmc > gpurun_out/$TAG/pmc_run.txt 2>&1
cat gpurun_out/$TAG/pmc/summary.txt | cut -c1-160
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o step -- python $R/tools/nar_step_bench.py "M5_NAR_DUAL=1" > $R/gpurun_out/$TAG/step_prof_run.txt 2>&1
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv && grep -E "nar_sample|nar_uniform|Name" "$f" < /dev/null | cut -c1-150
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
timeout 300 python tools/nar_step_bench.py "M5_NAR_CPLAN=1" "M5_NAR_CPLAN=0" 2>&1 | tail -4 | tee gpurun_out/$TAG/cplan_ab.txt
