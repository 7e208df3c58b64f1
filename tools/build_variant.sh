#!/bin/bash
# A second tools library that differs from libmars5_hip_tools.so in ONE source file compiled with extra flags (same-box A/B of a
# kernel variant through M5_HIP_TOOLS_LIB).  usage: tools/build_variant.sh <name> <source stem> "<extra flags>"
set -e
cd "$(dirname "$0")/../mars5-tts_amd/csrc"
NAME=$1; SRC=$2; EXTRA=$3
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -DM5_TOOLS"
mkdir -p obj_tools_$NAME
cp obj_tools/*.o obj_tools_$NAME/
hipcc $FLAGS $EXTRA -c $SRC.hip -o obj_tools_$NAME/$SRC.o
hipcc --offload-arch=gfx950 -shared -fPIC obj_tools_$NAME/*.o -o ../libmars5_hip_tools_$NAME.so
echo "built $(cd .. && pwd)/libmars5_hip_tools_$NAME.so"
