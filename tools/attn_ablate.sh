#!/bin/bash
mkdir -p gpurun_out/ablate
for v in "M5_ATTN_VARIANT=0" "M5_ATTN_VARIANT=1" "M5_ATTN_VARIANT=2" "M5_ATTN_VARIANT=3" "M5_ATTN_NW=2" "M5_ATTN_NW=8"; do
  echo "== $v"; env $v python tools/attn_bench.py 2>&1 | grep -E "nar self  |nar cross|spk enc"
done
