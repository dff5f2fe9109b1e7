#!/bin/bash
# A/B helper: run the matching micro-benchmark once per library variant in tmp_ab/
# usage: benchmarks/ab_libs.sh "<args of matching_bench.py>" ...
for lib in tmp_ab/*.so; do
  for args in "$@"; do
    echo "== $lib :: $args"
    GVAE_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 benchmarks/matching_bench.py $args | grep -E "mpm_stream wall" || exit 1
  done
done
