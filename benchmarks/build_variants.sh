#!/bin/bash
# Diagnostic builds of the library next to the product build (tmp_diag/, not shipped):
#   tune.so   -DGVAE_TUNE          launch shape / item thresholds from the environment
#   trace.so  + -DGVAE_STREAM_TRACE per-item timeline records (benchmarks/stream_trace.py)
#   stamps.so + -DGVAE_STREAM_STAMPS per-phase cycle totals printed by a few work-groups
#   fstamps.so -DGVAE_FUSED_STAMPS  per-phase cycle totals of the S-free MPM, printed by the first graph
# extra compiler flags for every variant: $GVAE_EXTRA_FLAGS
set -e
cd "$(dirname "$0")/../graph-vae_amd/csrc"
make -s
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -w $GVAE_EXTRA_FLAGS"
mkdir -p ../../tmp_diag /tmp/gvae_var
for v in ${1:-tune trace stamps}; do
  case $v in
    tune) D="-DGVAE_TUNE";;
    trace) D="-DGVAE_TUNE -DGVAE_STREAM_TRACE";;
    stamps) D="-DGVAE_TUNE -DGVAE_STREAM_STAMPS";;
    fstamps) D="-DGVAE_FUSED_STAMPS";;
  esac
  /opt/rocm/bin/hipcc $F $D -c matching.hip -o /tmp/gvae_var/matching_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_diag/$v.so build/common.o build/dense.o build/lsap.o build/step.o /tmp/gvae_var/matching_$v.o
done
ls -la ../../tmp_diag
