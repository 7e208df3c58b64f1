# round 6, call u: residual-class GEMM with loader waves (cfg 12) against cfg 7: per shape, with its error, and inside the NAR step
exec < /dev/null
TAG=r6u; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
DLN=1 ONLY="nar out_proj,nar p.b,nar linear2,big out_proj,big linear2" SWEEP=7,12,7,12 timeout 400 python tools/gemm_bench.py > gpurun_out/$TAG/gemm_cfg12_dln.txt 2>&1
cut -c1-170 gpurun_out/$TAG/gemm_cfg12_dln.txt
ONLY="nar out_proj,nar p.b,nar linear2" SWEEP=7,12 timeout 400 python tools/gemm_bench.py > gpurun_out/$TAG/gemm_cfg12_plain.txt 2>&1
cut -c1-170 gpurun_out/$TAG/gemm_cfg12_plain.txt
M5_HIP_TOOLS=1 M5_GEMM_CFG_E2=12 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_epilogues or deferred_layernorm or row_tile" 2>&1 | tail -5 > gpurun_out/$TAG/tests_gemm_cfg12.txt
cat gpurun_out/$TAG/tests_gemm_cfg12.txt
timeout 500 python tools/nar_step_bench.py "M5_GEMM_CFG_E2=7" "M5_GEMM_CFG_E2=12" 2>&1 | tail -4 > gpurun_out/$TAG/nar_step_cfg12.txt
cat gpurun_out/$TAG/nar_step_cfg12.txt
