# round 6, call g: phase timeline of the persistent decode step after the Wo move; two more weight-request placements
exec < /dev/null
TAG=r6g; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 300 python tools/ar_mega_clock.py 2>&1 | grep -v amdgpu.ids | head -40 > gpurun_out/$TAG/ar_mega_clock.txt
head -22 gpurun_out/$TAG/ar_mega_clock.txt | cut -c1-100
for rep in 1 2; do for lib in libmars5_hip_tools.so libmars5_hip_tools_w13once.so libmars5_hip_tools_w2late.so; do
  echo "== $lib" >> gpurun_out/$TAG/ar_dma_placement_ab2.txt
  M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/$lib timeout 300 python tools/ar_step_bench.py "M5_AR_MEGA=1" 2>&1 | grep "round 1" | cut -c1-160 >> gpurun_out/$TAG/ar_dma_placement_ab2.txt
done; done
cat gpurun_out/$TAG/ar_dma_placement_ab2.txt
