# streaming rate by graph size and work-group size (tuning build: GVAE_MPM_NT)
for nt in 256 512 1024; do
for cfg in "33 512" "40 256" "48 256" "52 128" "56 128" "57 128" "60 128" "64 128" "70 64" "75 64" "82 32" "96 32"; do
  set -- $cfg
  echo -n "nt=$nt "
  GVAE_HIP_LIB=$PWD/tmp_diag/tune.so GVAE_MPM_NT=$nt timeout -k 10 120 python3 benchmarks/matching_bench.py $1 $2 12 0 1 2>/dev/null | grep "mpm_stream N" | sed 's/iter kernels.*executed/executed/'
done; done
