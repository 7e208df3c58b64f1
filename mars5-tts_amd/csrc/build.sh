#!/bin/bash
# Build the gfx950 libraries (cross-compiles without a GPU).  Usage: csrc/build.sh [--no-tools]
#   libmars5_hip.so        the product: no environment knobs, no diagnostics, no ablation kernels
#   libmars5_hip_tools.so  the same sources with -DM5_TOOLS: tuning sweeps / A-B knobs / probes for tools/*.py
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result"
SRCS="gemm gemm16 gemm_skinny attention attention16 xattn_absorb rowops ar_decode ar_mega ar_batch nar_sample util stage_plan"
build_one() {   # $1 = object dir, $2 = extra flags, $3 = output library
  mkdir -p $1
  pids=()
  for f in $SRCS; do
    stale=0
    for h in *.h ../../include/mars5_hip.h; do [ $h -nt $1/$f.o ] && stale=1; done
    if [ ! -f $1/$f.o ] || [ $f.hip -nt $1/$f.o ] || [ $stale = 1 ]; then
      hipcc $FLAGS $2 -c $f.hip -o $1/$f.o &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait $p; done
  hipcc --offload-arch=gfx950 -shared -fPIC $1/*.o -o $3
  echo "built $(cd .. && pwd)/$(basename $3)"
}
build_one obj "" ../libmars5_hip.so
if [ "$1" != "--no-tools" ]; then
  build_one obj_tools "-DM5_TOOLS" ../libmars5_hip_tools.so
fi
