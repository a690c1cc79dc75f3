#!/bin/bash
# Build A/B variants of libdsaspp.so that differ in ONE source file's -D flags:
#   tools/ab_build.sh dgrad_group "-DDGG_ABL=1" abl1   ->  ab/libdsaspp_abl1.so
# (run the product build first; measure with DSASPP_LIB_PATH=ab/libdsaspp_abl1.so)
set -e
cd "$(dirname "$0")/.."
src=$1; flags=$2; tag=$3
C=semantic-segmentation_amd/csrc
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function $flags -c $C/$src.hip -o ab/${src}_$tag.o
objs=""
for f in $C/*.o; do
  b=$(basename $f .o)
  if [ "$b" = "$src" ]; then objs="$objs ab/${src}_$tag.o"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libdsaspp_$tag.so $objs
echo ab/libdsaspp_$tag.so
