# round 6, call p: repeatability of attn16w_kernel at large grids (an intermittent wrong result was seen once at B=16)
exec < /dev/null
TAG=r6p; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export CASES="16,16,2240,2240;2,16,1349,1349;4,16,1349,1349;2,16,5399,5399" STRESS=60
for rep in 1 2 3; do for v in 0 3 5; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_stress.txt
cut -c1-150 gpurun_out/$TAG/attn_stress.txt
