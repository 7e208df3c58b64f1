# round 6, call ae: where is the GPU idle during one bench utterance? (kernel trace of 3 utterances, gaps of the last)
exec < /dev/null
TAG=r6ae; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/prof -o utt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-batch-leg --no-extra-legs > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py "$f" > gpurun_out/$TAG/gaps.txt 2>&1; cat gpurun_out/$TAG/gaps.txt | cut -c1-200
rm -rf gpurun_out/$TAG/prof
