#!/bin/bash
# Developer tool: an experimental build of the library with extra -D flags on ONE source file, linked with the release objects.
#   tools/build_variant.sh <name> <source.hip> "<flags>"   ->  csrc/libaoc_hip_<name>.so   (load with AOC_LIB_FILE=libaoc_hip_<name>.so)
set -e
name=$1; src=$2; flags=$3
C=$(dirname "$0")/../robust-video-object-segmentation_amd/csrc
make -s -j4 -C $C
mkdir -p $C/build_var
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $flags -c $C/$src -o $C/build_var/${name}_${src%.hip}.o
objs=""
for f in $C/build/*.o; do b=$(basename $f); if [ "$b" == "${src%.hip}.o" ]; then objs="$objs $C/build_var/${name}_${src%.hip}.o"; else objs="$objs $f"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $C/libaoc_hip_${name}.so
echo built libaoc_hip_${name}.so
