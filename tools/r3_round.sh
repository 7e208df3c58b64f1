#!/bin/bash
# Round-3 GPU-box runner.  usage: tools/r3_round.sh <tag> <step>...   (outputs under gpurun_out/<tag>/)
# narbase / arbase / narprof A/B against the PREVIOUS round's library: build it from that round's tree first, e.g.
#   git worktree add /tmp/base <round-2 commit> && bash /tmp/base/mars5-tts_amd/csrc/build.sh &&
#   cp /tmp/base/mars5-tts_amd/libmars5_hip_tools.so mars5-tts_amd/libmars5_hip_tools_base.so   (untracked; travels with gpurun)
TAG=${1:-r3}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# a box whose GPU faults on the first copy (seen once in round 3: every process died in its first H2D transfer and the last
# leg then sat in its timeout) must not eat the round's GPU minutes
if ! timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum().cpu()) == float(2 << 20)" > $OUT/sanity.log 2>&1; then
  echo "GPU sanity check failed on this box: giving up"; tail -3 $OUT/sanity.log; exit 3
fi
for s in "$@"; do
  case $s in
    test) timeout 1700 python -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3 ;;
    parity) timeout 1200 python -m pytest tests/test_gpu_parity16.py -m gpu -q -s --durations=10 > $OUT/parity.log 2>&1; echo "parity rc=$?"; grep -E "^AR |^NAR |^nar_sample|passed|failed|Error|assert" $OUT/parity.log | tail -40 ;;
    yard) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/yard -o y -- python tools/blas_yardstick.py > $OUT/yard.log 2>&1; echo "yard rc=$?"; cat $OUT/yard.log | grep '^{'
          find $OUT/yard -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {} | cut -c1-260' > $OUT/yard_kernel_stats_head.txt
          find $OUT/yard -name "*kernel_trace.csv" -size +20M -delete ;;
    abl) for a in 0 1 2 3; do echo "== M5_GEMM_ABL=$a"; M5_GEMM_ABL=$a ONLY="nar self qkv,nar out_proj,nar swiglu,nar linear2" timeout 300 python tools/gemm_bench.py 2>&1 | grep "nar "; done > $OUT/gemm_abl.log 2>&1; echo "abl rc=$?"; cat $OUT/gemm_abl.log ;;
    attnq) CASES="${ATTN_CASES:-1,16,1349,1349;2,16,1024,1349;2,16,1349,1349;2,16,1536,1349;2,16,2048,1349;2,16,3072,1349;4,16,1349,1349}" timeout 300 python tools/attn_bench.py > $OUT/attn_grid.log 2>&1; echo "attnq rc=$?"; cat $OUT/attn_grid.log ;;
    narbase) M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/libmars5_hip_tools_base.so timeout 600 python tools/nar_step_bench.py "BASE=r2" > $OUT/narbase.log 2>&1; echo "narbase rc=$?"; grep round $OUT/narbase.log
             timeout 600 python tools/nar_step_bench.py "CUR=r3" > $OUT/narcur.log 2>&1; grep round $OUT/narcur.log ;;
    arbase) M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/libmars5_hip_tools_base.so timeout 600 python tools/ar_step_bench.py "BASE=r2" > $OUT/arbase.log 2>&1; echo "arbase rc=$?"; grep round $OUT/arbase.log
            timeout 600 python tools/ar_step_bench.py "CUR=r3" > $OUT/arcur.log 2>&1; grep round $OUT/arcur.log ;;
    narprof) export TMPDIR=/tmp
             M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/libmars5_hip_tools_base.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pbase -o b -- python tools/nar_step_bench.py "BASE=r2" > $OUT/narprof_base.log 2>&1
             timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pcur -o c -- python tools/nar_step_bench.py "CUR=r3" > $OUT/narprof_cur.log 2>&1
             grep round $OUT/narprof_base.log $OUT/narprof_cur.log
             find $OUT -name "*kernel_trace.csv" -delete ;;
    narab) timeout 900 python tools/nar_step_bench.py ${NARAB} > $OUT/narab.log 2>&1; echo "narab rc=$?"; grep round $OUT/narab.log ;;
    arab) timeout 900 python tools/ar_step_bench.py ${ARAB} > $OUT/arab.log 2>&1; echo "arab rc=$?"; grep round $OUT/arab.log ;;
    c3) timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$?"; cat $OUT/c3.json; tail -2 $OUT/c3.err ;;
    c5) timeout 900 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-batch-leg > $OUT/c5.json 2> $OUT/c5.err; echo "c5 rc=$?"; cat $OUT/c5.json | cut -c1-1500; tail -2 $OUT/c5.err ;;
    bench) timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -2 $OUT/bench.err ;;
    benchq) timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-batch-leg > $OUT/benchq.json 2> $OUT/benchq.err; echo "benchq rc=$?"; cat $OUT/benchq.json; tail -3 $OUT/benchq.err ;;
    prof) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-batch-leg > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
          find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {} | cut -c1-200'
          find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete ;;
    gemmtest) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or absorbed" > $OUT/gemmtest.log 2>&1; echo "gemmtest rc=$?"; tail -5 $OUT/gemmtest.log ;;
    gemmab) for pf in 0 1; do echo "== M5_GEMM_PF=$pf"; M5_GEMM_PF=$pf ONLY="nar out_proj,nar linear2,nar head (1 of 7),big out_proj,big linear2" timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "nar |big "; done > $OUT/gemm_pf_ab.log 2>&1
            echo "gemmab rc=$?"; cat $OUT/gemm_pf_ab.log ;;
    conc) timeout 600 python tools/nar_concurrent_probe.py > $OUT/conc.log 2>&1; echo "conc rc=$?"; grep -E "round|Error" $OUT/conc.log ;;
    l0qkv) SWEEP=${SWEEP:-0,1,2,3,4,7} ONLY="nar l0 qkv" timeout 300 python tools/gemm_bench.py > $OUT/l0qkv.log 2>&1; echo "l0qkv rc=$?"; grep "nar " $OUT/l0qkv.log ;;
    concb) CONC=1 MIXED=${MIXED:-1} timeout 900 python tools/nar_batch_bench.py ${CONCB:-4 8} > $OUT/concb.log 2>&1; echo "concb rc=$?"; grep "U=" $OUT/concb.log ;;
    pmcres) NOATTN=1 ONLY="nar out_proj,nar linear2" bash tools/pmc_traffic.sh $TAG/pmc > $OUT/pmcres.log 2>&1; echo "pmcres rc=$?"; cat $OUT/pmc/summary.txt | head -60 ;;
    attnab) for kh in 1 2; do echo "== M5_ATTN_KH=$kh"; M5_ATTN_KH=$kh timeout 300 python tools/attn_bench.py 2>&1 | grep -E "nar|spk|small|causal"
              M5_ATTN_KH=$kh CASES="16,16,2240,2240;2,16,5399,5399;1,16,1349,1349;4,16,1349,1349" timeout 300 python tools/attn_bench.py 2>&1 | grep case; done > $OUT/attn_kh_ab.log 2>&1
            echo "attnab rc=$?"; cat $OUT/attn_kh_ab.log
            timeout 600 python tools/nar_step_bench.py "M5_ATTN_KH=1" "M5_ATTN_KH=2" > $OUT/nar_kh_ab.log 2>&1; grep round $OUT/nar_kh_ab.log
            timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" > $OUT/attntest.log 2>&1; echo "attntest rc=$?"; tail -3 $OUT/attntest.log ;;
    pmcgemm) SWEEP=-1 ONLY="${PMC_ONLY:-nar out_proj,nar linear2,nar swiglu}" bash tools/pmc_gemm.sh $TAG/pmcg > $OUT/pmcgemm.log 2>&1; echo "pmcgemm rc=$?"
             python tools/pmc_summary.py $OUT/pmcg > $OUT/pmcg_summary.txt; find $OUT/pmcg -name "*.csv" -size +3M -delete; head -120 $OUT/pmcg_summary.txt ;;
    c3fly) for f in ${C3FLY:-1 2 3}; do timeout 600 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --nar-in-flight $f > $OUT/c3_fly$f.json 2> $OUT/c3_fly$f.err; echo "c3 in-flight $f rc=$?"
             python -c "import json,sys; d=json.load(open('$OUT/c3_fly$f.json')); print(d['value'], d['time_split_s_per_step'])"; done ;;
    smoke) timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log ;;
    *) echo "unknown step $s" ;;
  esac
done
