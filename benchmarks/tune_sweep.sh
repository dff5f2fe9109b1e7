#!/bin/bash
# Sweep of the streamed-MPM launch shape on a tuning build (-DGVAE_TUNE) of the library.
# usage: benchmarks/tune_sweep.sh <lib.so> "<nt values>" "<high values>" "<low values>" "<bench args>" ...
lib=$1; nts=$2; highs=$3; lows=$4; shift 4
for args in "$@"; do
  for nt in $nts; do for high in $highs; do for low in $lows; do
    r=$(GVAE_HIP_LIB=$PWD/$lib GVAE_MPM_NT=$nt GVAE_MPM_HIGH=$high GVAE_MPM_LOW=$low timeout -k 10 120 python3 benchmarks/matching_bench.py $args | grep -E "mpm_stream wall|TB/s algorithmic" | sed -e 's/.*executed graph-iters//' -e 's/mpm_stream wall//' | tr '\n' ' ') || exit 1
    echo "[$args] nt=$nt high=$high low=$low :: $r"
  done; done; done
done
