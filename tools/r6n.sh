exec < /dev/null
TAG=r6n; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export CASES="1,1,128,64;1,1,128,100;1,1,128,128;1,1,128,192;1,1,128,256;1,2,300,1349"
for v in 0 5 6; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/$TAG/attn_w_debug.txt
cut -c1-150 gpurun_out/$TAG/attn_w_debug.txt
