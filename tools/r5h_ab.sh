# whole-utterance A/B of one host-engine knob on one box, alternating (tools library: both sides run the same kernels)
exec < /dev/null      # nothing in a gpurun script may wait on stdin (a `head` without a file name once held a box until the call limit)
# usage: bash tools/r5h_ab.sh KNOB [tag]
KNOB=${1:-M5_NAR_PHILOX}; TAG=${2:-r5h}
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for rep in 1 2; do for P in 0 1; do
  env M5_HIP_TOOLS=1 $KNOB=$P timeout 200 python bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-batch-leg --no-preflight 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['per_step']['steps']
print('$KNOB=$P rep $rep value', d['value'], 'wall', [x[0] for x in s], 'nar_ms', [x[2] for x in s][:3], 'ar_ms', [x[1] for x in s][:3])" >> gpurun_out/$TAG/ab.txt
done; done
cat gpurun_out/$TAG/ab.txt
