# round 6, call ac: lazy running maximum in the attention kernels (skip the rescale while a tile's max exceeds the reference by <= 2^8)
exec < /dev/null
TAG=r6ac; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
V=$PWD/mars5-tts_amd/libmars5_hip_tools_nolazy.so
for r in 0 1; do
  CASES="2,16,1349,1349;1,16,1349,1349;2,16,5399,5399;8,16,1500,1500" REP=2 timeout 300 python tools/attn_bench.py 2>&1 | grep "case" | sed "s/^/lazy   /" >> gpurun_out/$TAG/attn_cases.txt
  M5_HIP_TOOLS_LIB=$V CASES="2,16,1349,1349;1,16,1349,1349;2,16,5399,5399;8,16,1500,1500" REP=2 timeout 300 python tools/attn_bench.py 2>&1 | grep "case" | sed "s/^/nolazy /" >> gpurun_out/$TAG/attn_cases.txt
done
cut -c1-150 gpurun_out/$TAG/attn_cases.txt
for r in 0 1; do
  timeout 300 python tools/nar_step_bench.py "X=lazy" 2>&1 | grep round >> gpurun_out/$TAG/nar_step_ab.txt
  M5_HIP_TOOLS_LIB=$V timeout 300 python tools/nar_step_bench.py "X=nolazy" 2>&1 | grep round >> gpurun_out/$TAG/nar_step_ab.txt
done
cat gpurun_out/$TAG/nar_step_ab.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity16.py -m gpu -q -k "attention or attn or nar_full_size or nar_tiny or parity or batch" 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/$TAG/tests.txt
cat gpurun_out/$TAG/tests.txt
