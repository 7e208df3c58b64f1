# round 6, call l: attention time against the number of key tiles (fixed cost vs per-tile cost), one and two workgroups per CU
exec < /dev/null
TAG=r6l; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export CASES="1,16,1349,64;1,16,1349,320;1,16,1349,704;1,16,1349,1408;1,16,1349,2816;2,16,1349,64;2,16,1349,320;2,16,1349,704;2,16,1349,1408;2,16,1349,2816;2,16,1024,64;2,16,1024,1408;2,16,1024,2816"
for v in 0 2; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/$TAG/attn_tiles.txt
cut -c1-130 gpurun_out/$TAG/attn_tiles.txt
export M5_ATTN_SCHED=2 CASES="1,16,1349,1349;2,16,1349,1349;16,16,2240,2240"
for rep in 1 2; do for v in 0 6; do echo "== M5_ATTN_ABL=$v"; M5_ATTN_ABL=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_dma_in_phase2.txt
cut -c1-130 gpurun_out/$TAG/attn_dma_in_phase2.txt
