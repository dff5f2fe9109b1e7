set -e
mkdir -p gpurun_out
out=gpurun_out/final_sizes.txt
: > $out
echo "== sizes new" >> $out
bash benchmarks/stream_sizes.sh >> $out 2>&1
for cfg in "88 32" "91 32" "99 32" "104 32" "112 16" "120 16" "128 16"; do
  set -- $cfg
  timeout -k 10 120 python3 benchmarks/matching_bench.py $1 $2 12 0 1 2>/dev/null | grep "mpm_stream N" | sed 's/iter kernels.*executed/executed/' >> $out
done
echo "== ragged new" >> $out
timeout -k 10 300 python -u benchmarks/ragged_throughput.py 64 12 0 1 2 3 5 8 9 14 18 23 >> $out 2>&1
