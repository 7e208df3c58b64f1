# round 6, call af: GRAPH_GROUP decode steps per replayed hipGraph against one
exec < /dev/null
TAG=r6af; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 600 python tools/ar_step_bench.py "M5_AR_GROUP=1" "M5_AR_GROUP=8" "M5_AR_GROUP=16" "M5_AR_GROUP=32" 2>&1 | grep round > gpurun_out/$TAG/ar_group.txt; cat gpurun_out/$TAG/ar_group.txt
timeout 1200 python -m pytest tests/test_gpu_parity16.py tests/test_gpu_e2e.py -m gpu -q -k "ar_ or persistent or tts or window or eos" 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/$TAG/tests.txt; cat gpurun_out/$TAG/tests.txt
