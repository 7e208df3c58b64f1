# round 6, call o: loader-wave attention (attn16w_kernel): 3 computing waves + loader (96-row workgroups, sched 3), 4 + loader at 168 VGPRs (spills, sched 4), 4 + loader one workgroup per CU (sched 5)
exec < /dev/null
TAG=r6o; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for rep in 1 2; do for v in 0 3 4 5; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_loader_wave_cases.txt
cut -c1-150 gpurun_out/$TAG/attn_loader_wave_cases.txt | head -34
export CASES="1,16,1349,1349;2,16,1349,1349;4,16,1349,1349;16,16,2240,2240;2,16,5399,5399"
for rep in 1 2; do for v in 0 3 5; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_loader_wave.txt
cut -c1-150 gpurun_out/$TAG/attn_loader_wave.txt
