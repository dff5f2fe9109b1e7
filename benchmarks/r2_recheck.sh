set -e
O=gpurun_out/r2q; mkdir -p $O
timeout -k 10 900 python3 -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --workload c1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err
timeout -k 10 300 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 400 python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 400 python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
echo done
