# round 6, call f: where the persistent decode step requests its next weight sets (same arithmetic; two tools builds per variant, alternated)
exec < /dev/null
TAG=r6f; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for rep in 1 2; do for lib in libmars5_hip_tools.so libmars5_hip_tools_wolate.so libmars5_hip_tools_wqearly.so libmars5_hip_tools_both.so; do
  echo "== $lib" >> gpurun_out/$TAG/ar_dma_placement_ab.txt
  M5_HIP_TOOLS_LIB=$PWD/mars5-tts_amd/$lib timeout 300 python tools/ar_step_bench.py "M5_AR_MEGA=1" 2>&1 | grep round | cut -c1-160 >> gpurun_out/$TAG/ar_dma_placement_ab.txt
done; done
cat gpurun_out/$TAG/ar_dma_placement_ab.txt
