# round 6, call i: hand-scheduled attention kernel (attn16s_kernel) against attn16_kernel, bitwise (csum) and timed; AR weight-request placement on the product path
exec < /dev/null
TAG=r6i; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1 2; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_sched.txt
cut -c1-170 gpurun_out/$TAG/attn_sched.txt | head -48
for v in 0 2; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v CASES="1,16,1349,1349;2,16,1024,1349;2,16,1349,1349;4,16,1349,1349;16,16,2240,2240;2,16,5399,5399" timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/$TAG/attn_sched_grid.txt
cut -c1-170 gpurun_out/$TAG/attn_sched_grid.txt
M5_HIP_TOOLS=1 M5_ATTN_SCHED=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention" 2>&1 | tail -5 > gpurun_out/$TAG/tests_attention_sched2.txt
cat gpurun_out/$TAG/tests_attention_sched2.txt
timeout 900 python -m pytest tests/test_gpu_parity16.py -m gpu -q -k "persistent" 2>&1 | tail -5 > gpurun_out/$TAG/tests_ar_persistent.txt
cat gpurun_out/$TAG/tests_ar_persistent.txt
timeout 300 python tools/ar_step_bench.py "M5_AR_MEGA=1" 2>&1 | grep "round" | cut -c1-160 > gpurun_out/$TAG/ar_step.txt
cat gpurun_out/$TAG/ar_step.txt
for v in 0 2; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 500 python tools/nar_step_bench.py "M5_NAR_DUAL=1" 2>&1 | tail -3; done > gpurun_out/$TAG/nar_step_attn_sched.txt
cat gpurun_out/$TAG/nar_step_attn_sched.txt
