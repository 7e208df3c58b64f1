# round 6, call aa: three-stage 128x256 tile (cfg 13, tools build) against the two-stage 192x384 / 192x192 regions at large M
exec < /dev/null
TAG=r6aa; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
ONLY="big swiglu,big qkv,nar swiglu,nar self qkv" SWEEP=5,2,13,5,2,13 timeout 600 python tools/gemm_bench.py > gpurun_out/$TAG/gemm_cfg13.txt 2>&1
grep cfg= gpurun_out/$TAG/gemm_cfg13.txt | cut -c1-150
