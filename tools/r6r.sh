# round 6, call r: loader-wave attention as the product's small-grid form: tests, NAR step A/B, cold-launch repeatability with the fixed harness
exec < /dev/null
TAG=r6r; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention" 2>&1 | tail -4 > gpurun_out/$TAG/tests_attention.txt
cat gpurun_out/$TAG/tests_attention.txt
for v in 0 3 0 3; do echo "== M5_ATTN_SCHED=$v"; M5_ATTN_SCHED=$v timeout 500 python tools/nar_step_bench.py "M5_NAR_DUAL=1" 2>&1 | tail -2; done > gpurun_out/$TAG/nar_step_attn_loader_ab.txt
cat gpurun_out/$TAG/nar_step_attn_loader_ab.txt
export CASES="1,16,1349,1349;2,16,1349,1349;4,16,1349,1349;16,16,2240,2240" REP=1
for i in $(seq 1 10); do for v in 3; do M5_ATTN_SCHED=$v timeout 100 python tools/attn_bench.py 2>&1 | grep "16,16,2240" | sed "s/^/sched $v run $i: /"; done; done > gpurun_out/$TAG/attn_cold.txt
awk '{print $1,$2,$(NF-1),$NF}' gpurun_out/$TAG/attn_cold.txt | sort | uniq -c
unset CASES REP
timeout 1500 python -m pytest tests/test_gpu_parity16.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/$TAG/tests_parity16.txt
cat gpurun_out/$TAG/tests_parity16.txt
