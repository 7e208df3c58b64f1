exec < /dev/null
TAG=r6h; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for rep in 1 2 3; do for lib in libmars5_hip_tools.so libmars5_hip_tools_w13once.so libmars5_hip_tools_once_w2late.so; do
  echo "== $lib" >> gpurun_out/$TAG/ar_dma_placement_ab3.txt
  M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/$lib timeout 300 python tools/ar_step_bench.py "M5_AR_MEGA=1" 2>&1 | grep "round 1" | cut -c1-160 >> gpurun_out/$TAG/ar_dma_placement_ab3.txt
done; done
cat gpurun_out/$TAG/ar_dma_placement_ab3.txt
