# round 6, call q: cold first launches of attn16w_kernel at the batched shape (fresh process each time) -- hunting one intermittent wrong result
exec < /dev/null
TAG=r6q; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export CASES="1,16,1349,1349;2,16,1349,1349;4,16,1349,1349;16,16,2240,2240" REP=1
for i in $(seq 1 24); do for v in 3 5; do M5_ATTN_SCHED=$v timeout 100 python tools/attn_bench.py 2>&1 | grep "16,16,2240" | sed "s/^/sched $v run $i: /"; done; done > gpurun_out/$TAG/attn_cold.txt
cut -c1-170 gpurun_out/$TAG/attn_cold.txt | awk '{print $1,$2,$3,$4,$(NF-1),$NF}' | sort | uniq -c
