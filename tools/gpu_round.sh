#!/bin/bash
# One GPU-box round: parity tests, GEMM microbench (A/B), end-to-end bench, rocprofv3 kernel stats.
# usage: tools/gpu_round.sh <tag> [steps...]   steps default: test gemm bench prof
TAG=${1:-r}; shift
STEPS=${@:-test gemm bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for s in $STEPS; do
  case $s in
    test) timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log ;;
    sweep) SWEEP=${SWEEP:-0,1,2,3,4,5,6,7} timeout 600 python tools/gemm_bench.py > $OUT/gemm_sweep.log 2>&1; echo "sweep rc=$?"; cat $OUT/gemm_sweep.log ;;
    attn) timeout 300 python tools/attn_bench.py > $OUT/attn_v2.log 2>&1; echo "attn v2 rc=$?"; tail -12 $OUT/attn_v2.log
          M5_ATTN_V1=1 timeout 300 python tools/attn_bench.py > $OUT/attn_v1.log 2>&1; echo "attn v1 rc=$?"; tail -8 $OUT/attn_v1.log ;;
    gemm) timeout 300 python tools/gemm_bench.py > $OUT/gemm_v2.log 2>&1; echo "gemm v2 rc=$?"; cat $OUT/gemm_v2.log | tail -25
          M5_GEMM_V1=1 timeout 300 python tools/gemm_bench.py > $OUT/gemm_v1.log 2>&1; echo "gemm v1 rc=$?"; tail -25 $OUT/gemm_v1.log ;;
    bench) timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err ;;
    benchfull) timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err ;;
    prof) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
          find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'
          find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete ;;
  esac
done
