#!/bin/bash
# Build libmars5_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [-j]
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
mkdir -p obj
pids=()
for f in gemm gemm16 gemm_skinny attention attention16 rowops ar_decode ar_batch nar_sample util; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ ../../include/mars5_hip.h -nt obj/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/*.o -o ../libmars5_hip.so
echo "built $(cd .. && pwd)/libmars5_hip.so"
