# round 6, call m: loader-wave attention kernel (attn16w_kernel) against attn16_kernel: bitwise (csum) and timed
exec < /dev/null
TAG=r6m; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export CASES="1,16,1349,1349;2,16,1349,1349;2,16,1349,64;2,16,1349,100;4,16,1349,1349;16,16,2240,2240"
for rep in 1 2; do for v in 0 4 5; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_loader_wave.txt
cut -c1-150 gpurun_out/$TAG/attn_loader_wave.txt
unset CASES
for v in 0 4; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/$TAG/attn_loader_wave_cases.txt
cut -c1-150 gpurun_out/$TAG/attn_loader_wave_cases.txt
