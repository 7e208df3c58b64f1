# round 6, call j: VALU instruction-rate probe; SQ counters of the attention kernel at the NAR shape and at a batched shape
exec < /dev/null
TAG=r6j; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 120 tools/probes/valu_rate > gpurun_out/$TAG/valu_rate.txt 2>&1
cat gpurun_out/$TAG/valu_rate.txt
export M5_ATTN_SCHED=0 CASES="2,16,1349,1349;16,16,2240,2240" REP=2
bash tools/pmc_run.sh $TAG/pmc python tools/attn_bench.py > /dev/null 2>&1
cat gpurun_out/$TAG/pmc/summary.txt | cut -c1-150
