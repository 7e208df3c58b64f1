# round 6, call ag: STEP_GROUP reverse steps per replayed hipGraph against one; AR groups already on
exec < /dev/null
TAG=r6ag; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 600 python tools/nar_step_bench.py "M5_NAR_GROUP=0" "M5_NAR_GROUP=1" 2>&1 | grep round > gpurun_out/$TAG/nar_group.txt; cat gpurun_out/$TAG/nar_group.txt
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity16.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/$TAG/tests.txt; cat gpurun_out/$TAG/tests.txt
timeout 900 python bench.py --steps 8 --warmup 2 --no-extra-legs --no-batch-leg --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['time_split_ms'], d['nar_loop']['ms_per_step'], d['roofline_ar_decode']['us_per_token'])" | tee gpurun_out/$TAG/bench_short.txt
