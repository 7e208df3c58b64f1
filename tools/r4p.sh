export TMPDIR=/tmp
OUT=gpurun_out/r4p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity16.py -x -q -k "ar_ or persistent" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -k "ar or gemv or tts" 2>&1 | tail -4
M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/libmars5_hip_tools_base.so timeout 600 python tools/ar_step_bench.py "BASE=fmaf" > $OUT/arbase.log 2>&1; grep round $OUT/arbase.log
timeout 600 python tools/ar_step_bench.py "CUR=dot2" > $OUT/arcur.log 2>&1; grep round $OUT/arcur.log
