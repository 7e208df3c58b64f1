# round 6, call e: upper bound of a weight prefetch for the NAR step (every decoder layer on layer 0's weights: cache-resident, WRONG results)
exec < /dev/null
TAG=r6e; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 500 python tools/nar_step_bench.py "M5_NAR_SAMEW=0" "M5_NAR_SAMEW=1" 2>&1 | tail -4 | tee gpurun_out/$TAG/samew.txt
