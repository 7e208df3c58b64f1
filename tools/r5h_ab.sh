mkdir -p gpurun_out/r5h; export TMPDIR=/tmp
for rep in 1 2; do for P in 0 1; do
  M5_HIP_TOOLS=1 M5_NAR_PHILOX=$P timeout 200 python bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-batch-leg --no-preflight 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['per_step']['steps']
print('PHILOX=$P rep $rep value', d['value'], 'wall', [x[0] for x in s], 'nar_ms', [x[2] for x in s][:3], 'ar_ms', s[0][1])" >> gpurun_out/r5h/ab.txt
done; done
cat gpurun_out/r5h/ab.txt
