# steady-state rate of the streaming kernel by (uniform) graph size, padded layout, no early stop
for cfg in "16 4096" "20 2048" "24 1024" "28 1024" "30 512" "32 512" "33 512" "36 512" "40 256" "44 256" "47 256" "48 256" "52 128" "56 128" "57 128" "60 128" "64 128" "70 64" "75 64" "82 32" "96 32"; do
  set -- $cfg
  timeout -k 10 120 python3 benchmarks/matching_bench.py $1 $2 12 0 1 2>/dev/null | grep "mpm_stream N" | sed 's/iter kernels.*executed/executed/'
done
