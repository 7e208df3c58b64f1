# round 6, call k: cost attribution of the attention tile loop (ablations of attn16s_kernel; results WRONG except ABL 0 / 6)
exec < /dev/null
TAG=r6k; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export M5_ATTN_SCHED=2 CASES="1,16,1349,1349;2,16,1349,1349;16,16,2240,2240"
for rep in 1 2; do for v in 0 1 2 3 4 5 6 7; do echo "== M5_ATTN_ABL=$v"; M5_ATTN_ABL=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/$TAG/attn_ablation.txt
cut -c1-130 gpurun_out/$TAG/attn_ablation.txt
