# streaming rate by graph size with 16 lanes per row for rows that are not multiples of 128 bytes
# (tuning build: GVAE_MPM_WIDE = 0 off, 1 up to 48 vectors per row, 2 up to 16 vectors per row)
for wide in 0 1 2; do
for cfg in "33 512" "36 512" "40 256" "44 256" "47 256" "52 128" "57 128" "60 128" "70 64" "75 64" "82 32"; do
  set -- $cfg
  echo -n "wide=$wide "
  GVAE_HIP_LIB=$PWD/tmp_diag/tune.so GVAE_MPM_WIDE=$wide timeout -k 10 120 python3 benchmarks/matching_bench.py $1 $2 12 0 1 2>/dev/null | grep "mpm_stream N" | sed 's/iter kernels.*executed graph-iters [0-9]*;//'
done; done
