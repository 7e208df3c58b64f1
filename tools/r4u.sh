export TMPDIR=/tmp
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -x -q -k "batch or gemm or deferred" 2>&1 | tail -3
MIXED=1 timeout 300 python tools/nar_batch_bench.py 32 2>&1 | grep "U="
timeout 900 python bench.py --workload c3 --batch 32 --steps 1 --warmup 1 --no-preflight > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$? $(cut -c1-160 $OUT/c3.json)"
