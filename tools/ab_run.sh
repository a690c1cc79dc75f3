#!/bin/bash
# Run one micro-benchmark case against the product library and every ab/libdsaspp_*.so variant.
#   tools/ab_run.sh slab_multi4_16 16 [cin]
case=$1; batch=${2:-16}; cin=${3:-1248}
echo "== product"; timeout -k 10 120 python tools/conv_bench.py --only $case --batch $batch --cin $cin || exit 1
for so in ab/libdsaspp_*.so; do
  echo "== $so"; DSASPP_LIB_PATH=$PWD/$so timeout -k 10 120 python tools/conv_bench.py --only $case --batch $batch --cin $cin || exit 1
done
