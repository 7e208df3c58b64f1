# round 6, call s: SQ counters of the loader-wave attention kernel at the NAR shape
exec < /dev/null
TAG=r6s; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
export M5_ATTN_SCHED=3 CASES="1,16,1349,1349;2,16,1349,1349" REP=2
bash tools/pmc_run.sh $TAG/pmc python tools/attn_bench.py > /dev/null 2>&1
cat gpurun_out/$TAG/pmc/summary.txt | cut -c1-150
