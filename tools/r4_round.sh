#!/bin/bash
# Round-4 GPU-box runner.  usage: tools/r4_round.sh <tag> <step>...   (outputs under gpurun_out/<tag>/)
TAG=${1:-r4}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if ! timeout 180 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum().cpu()) == float(2 << 20)" > $OUT/sanity.log 2>&1; then
  echo "GPU sanity check failed on this box: giving up"; tail -3 $OUT/sanity.log; exit 3
fi
for s in "$@"; do
  case $s in
    test) timeout 1700 python -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3 ;;
    testall) timeout 1700 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -8 ;;
    driver) timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver.json 2> $OUT/driver.err; echo "driver rc=$?"; tail -1 $OUT/driver.json | cut -c1-1800; wc -l $OUT/driver.json; tail -3 $OUT/driver.err | cut -c1-300 ;;
    quick3) for i in 1 2 3; do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-parity --no-batch-leg > $OUT/quick$i.json 2> $OUT/quick$i.err; echo "quick$i rc=$? $(tail -1 $OUT/quick$i.json | cut -c1-160)"; grep -i "fault" $OUT/quick$i.err; done ;;
    guardtail) timeout ${GUARD_TIMEOUT:-900} python tools/guard_run.py --mode tail --log $OUT/guard_tail -- --steps 2 --warmup 1 --no-cpu-baseline --no-preflight ${GUARD_ARGS} > $OUT/guard_tail.json 2> $OUT/guard_tail.err; echo "guardtail rc=$?"
               tail -1 $OUT/guard_tail.json | cut -c1-600; grep -iE "fault|guard_run|Error" $OUT/guard_tail.err | head -8; head -3 $OUT/guard_tail/alloc.log; wc -l $OUT/guard_tail/alloc.log
               python tools/guard_report.py $OUT/guard_tail $OUT/guard_tail.err | head -20; gzip -f $OUT/guard_tail/alloc.log ;;
    guardhead) timeout ${GUARD_TIMEOUT:-900} python tools/guard_run.py --mode head --log $OUT/guard_head -- --steps 2 --warmup 1 --no-cpu-baseline --no-preflight ${GUARD_ARGS} > $OUT/guard_head.json 2> $OUT/guard_head.err; echo "guardhead rc=$?"
               tail -1 $OUT/guard_head.json | cut -c1-600; grep -iE "fault|guard_run|Error" $OUT/guard_head.err | head -8; wc -l $OUT/guard_head/alloc.log
               python tools/guard_report.py $OUT/guard_head $OUT/guard_head.err | head -20; gzip -f $OUT/guard_head/alloc.log ;;
    guardtrace) timeout ${GUARD_TIMEOUT:-900} python tools/guard_run.py --mode ${GUARD_MODE_:-tail} --trace --log $OUT/guard_trace -- --steps 1 --warmup 0 --no-cpu-baseline --no-preflight --no-roofline --no-parity --no-batch-leg ${GUARD_ARGS} > $OUT/guard_trace.json 2> $OUT/guard_trace.err; echo "guardtrace rc=$?"
               grep -iE "fault|guard_run|Error" $OUT/guard_trace.err | head -8; tail -5 $OUT/guard_trace/launch.log
               python tools/guard_report.py $OUT/guard_trace $OUT/guard_trace.err | head -20; gzip -f $OUT/guard_trace/alloc.log ;;
    prof) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-batch-leg --no-preflight > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
          find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$OUT'/kernel_stats.csv; head -30 {} | cut -c1-200'
          find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete ;;
    bench) timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.json | cut -c1-2500; tail -2 $OUT/bench.err ;;
    benchq) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-batch-leg > $OUT/benchq.json 2> $OUT/benchq.err; echo "benchq rc=$?"; tail -1 $OUT/benchq.json | cut -c1-3000; tail -3 $OUT/benchq.err ;;
    narab) timeout 900 python tools/nar_step_bench.py ${NARAB} > $OUT/narab.log 2>&1; echo "narab rc=$?"; grep round $OUT/narab.log ;;
    arab) timeout 900 python tools/ar_step_bench.py ${ARAB} > $OUT/arab.log 2>&1; echo "arab rc=$?"; grep round $OUT/arab.log ;;
    gemm) timeout 600 python tools/gemm_bench.py > $OUT/gemm.log 2>&1; echo "gemm rc=$?"; cat $OUT/gemm.log | tail -40 ;;
    yard) timeout 600 python tools/blas_yardstick.py > $OUT/yard.log 2>&1; echo "yard rc=$?"; tail -30 $OUT/yard.log ;;
    c3) timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$?"; cat $OUT/c3.json; tail -2 $OUT/c3.err ;;
    cmd) bash -c "$R4_CMD" > $OUT/cmd.log 2>&1; echo "cmd rc=$?"; tail -${R4_TAIL:-40} $OUT/cmd.log ;;
    *) echo "unknown step $s" ;;
  esac
done
