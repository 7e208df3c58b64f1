# round 6, call x: loader-wave residual GEMM with five stages (cfg 13, tools build) against cfg 12
exec < /dev/null
TAG=r6x; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
DLN=1 ONLY="nar out_proj,nar p.b,nar linear2,big linear2" SWEEP=12,13,12,13 timeout 400 python tools/gemm_bench.py > gpurun_out/$TAG/gemm_cfg13_dln.txt 2>&1
grep cfg= gpurun_out/$TAG/gemm_cfg13_dln.txt | cut -c1-150
timeout 500 python tools/nar_step_bench.py "M5_GEMM_CFG_E2=12" "M5_GEMM_CFG_E2=13" 2>&1 | tail -4 > gpurun_out/$TAG/nar_step_cfg13.txt
cat gpurun_out/$TAG/nar_step_cfg13.txt
