# round 6, call z: final tree -- whole GPU suite, the driver's bench command, one-utterance kernel stats, PMC traffic of the dominant class
exec < /dev/null
TAG=r6z; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$TAG/gpu_suite_full.txt
tail -8 gpurun_out/$TAG/gpu_suite_full.txt | cut -c1-250
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver.json 2> gpurun_out/$TAG/bench_driver.err
tail -c 3000 gpurun_out/$TAG/bench_driver.json | cut -c1-3000
tail -3 gpurun_out/$TAG/bench_driver.err
R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o utt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-batch-leg --no-extra-legs > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof_bench.err
cd $R; f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/utterance_kernel_stats.csv && head -8 "$f" | cut -c1-170
find gpurun_out/$TAG/prof -type f ! -name '*stats.csv' -delete
DLN=1 ONLY="nar out_proj,nar p.b,nar linear2" NOATTN=1 bash tools/pmc_traffic.sh $TAG/pmc > gpurun_out/$TAG/pmc_run.txt 2>&1
cat gpurun_out/$TAG/pmc/summary.txt | cut -c1-160
