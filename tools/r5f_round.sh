mkdir -p gpurun_out/r5f; export TMPDIR=/tmp
exec < /dev/null      # nothing in a gpurun script may wait on stdin (a `head` without a file name once held a box until the call limit)
for T in 64 128 256; do
  timeout 200 python bench.py --steps 1 --warmup 0 --no-roofline --no-parity --no-batch-leg --cpu-threads $T 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d.get('cpu_baseline',{}); print('threads', c.get('cores'), 'value', c.get('value'), c.get('sample','')[:260])" >> gpurun_out/r5f/cpu_threads.txt
done
cat gpurun_out/r5f/cpu_threads.txt
MROWS=35840 timeout 200 python tools/blas_yardstick.py 2>&1 | grep -v "WARNING\|amdgpu" > gpurun_out/r5f/yardstick_35840.txt; cat gpurun_out/r5f/yardstick_35840.txt
MROWS=98304 timeout 300 python tools/blas_yardstick.py 2>&1 | grep -v "WARNING\|amdgpu" > gpurun_out/r5f/yardstick_98304.txt; cat gpurun_out/r5f/yardstick_98304.txt
