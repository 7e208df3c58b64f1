#!/bin/bash
# Round-2 GPU-box runner.  usage: tools/r2_round.sh <tag> <step>...   (outputs under gpurun_out/<tag>/)
TAG=${1:-r2}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for s in "$@"; do
  case $s in
    test) timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3 ;;
    parity) timeout 1200 python -m pytest tests/test_gpu_parity16.py -m gpu -q -s --durations=10 > $OUT/parity.log 2>&1; echo "parity rc=$?"; grep -E "^AR |^NAR |^nar_sample|passed|failed|Error|assert" $OUT/parity.log | tail -30 ;;
    arpf) timeout 600 python tools/ar_step_bench.py ${ARPF:-"M5_AR_PREFETCH=0" "M5_AR_PREFETCH=2,M5_AR_PREFETCH_WGS=256" "M5_AR_PREFETCH=1,M5_AR_PREFETCH_WGS=256" "M5_AR_PREFETCH=2,M5_AR_PREFETCH_WGS=512" "M5_AR_PREFETCH=1,M5_AR_PREFETCH_WGS=512"} > $OUT/arpf.log 2>&1; echo "arpf rc=$?"; grep round $OUT/arpf.log ;;
    narab) timeout 600 python tools/nar_step_bench.py ${NARAB:-"M5_NAR_ABSORB=0" "M5_NAR_ABSORB=1" "M5_NAR_ABSORB=1,M5_XATTN_CFG=1" "M5_NAR_ABSORB=1,M5_XATTN_CFG=2"} > $OUT/narab.log 2>&1; echo "narab rc=$?"; grep round $OUT/narab.log ;;
    attn) bash tools/attn_ablate.sh > $OUT/attn_ablate.log 2>&1; echo "attn rc=$?"; grep -E "^==|nar self|nar cross" $OUT/attn_ablate.log ;;
    c3) timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$?"; cat $OUT/c3.json; tail -2 $OUT/c3.err ;;
    c5) timeout 900 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-batch-leg > $OUT/c5.json 2> $OUT/c5.err; echo "c5 rc=$?"; cat $OUT/c5.json | cut -c1-1500; tail -2 $OUT/c5.err ;;
    f16) timeout 600 python bench.py --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-batch-leg > $OUT/bench_f16.json 2> $OUT/bench_f16.err; echo "f16 rc=$?"; cat $OUT/bench_f16.json | cut -c1-700 ;;
    traffic) ONLY="nar swiglu" bash tools/pmc_traffic.sh $TAG/traffic > $OUT/traffic.log 2>&1; echo "traffic rc=$?"; cat gpurun_out/$TAG/traffic/summary.txt | grep -A3 -E "gemm16" | head -40 ;;
    c4) timeout 600 python bench.py --workload c4 --batch ${C4B:-4} --steps 1 --warmup 1 > $OUT/c4.json 2> $OUT/c4.err; echo "c4 rc=$?"; cat $OUT/c4.json; tail -2 $OUT/c4.err ;;
    bench) timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -2 $OUT/bench.err ;;
    benchq) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/benchq.json 2> $OUT/benchq.err; echo "bench rc=$?"; cat $OUT/benchq.json; tail -2 $OUT/benchq.err ;;
    prof) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-batch-leg > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
          find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {} | cut -c1-200'
          find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete ;;
    smoke) timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log ;;
    *) echo "unknown step $s" ;;
  esac
done
