# round 6, call c: the whole GPU suite on the build with stage plans + split-f16 fp32 mode, then the driver's bench command
exec < /dev/null
TAG=r6c; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$TAG/gpu_suite_full.txt
tail -15 gpurun_out/$TAG/gpu_suite_full.txt | cut -c1-250
grep -E "m5_nar_step|m5_ar_decode_step|RCCL|split-f16|known rows" gpurun_out/$TAG/gpu_suite_full.txt | cut -c1-220
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver.json 2> gpurun_out/$TAG/bench_driver.err
tail -c 6000 gpurun_out/$TAG/bench_driver.json
tail -5 gpurun_out/$TAG/bench_driver.err
