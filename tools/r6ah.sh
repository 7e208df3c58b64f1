# round 6, call ah: FINAL tree (AR decode steps grouped 8 per graph): whole GPU suite + the driver's bench command
exec < /dev/null
TAG=r6ah; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$TAG/gpu_suite_full.txt
tail -4 gpurun_out/$TAG/gpu_suite_full.txt | cut -c1-250
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver.json 2> gpurun_out/$TAG/bench_driver.err
python -c "
import json
d=json.loads(open('gpurun_out/$TAG/bench_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['time_split_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['nar_loop']['ms_per_step'], d['roofline_ar_decode']['us_per_token'], d['roofline_ar_decode']['frac'], d['c3_batch32']['value'], d['c5_longform']['value'], d['batch8_mixed_lengths']['value'], d['reference_precision']['value'], d['reference_precision_f16x3']['value'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
