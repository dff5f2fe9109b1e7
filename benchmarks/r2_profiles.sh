# round-2 measurement set: bench lines of every single-GPU workload, the c4 iteration sweep,
# PMC traffic of the streaming kernel at 128 nodes (FETCH_SIZE and WRITE_SIZE in separate passes)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p
mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --workload c1 > $O/bench_c1.json 2> $O/bench_c1.err
timeout -k 10 300 python3 bench.py --workload c3 --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err
for it in 50 100 300; do
  timeout -k 10 400 python3 bench.py --workload c4 --mpm-iters $it --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_iters$it.json 2> $O/bench_c4_iters$it.err
done
timeout -k 10 400 python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_c5_fetch -o f -- python3 benchmarks/matching_bench.py 128 16 50 0 1 > $O/pmc_c5_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_c5_write -o w -- python3 benchmarks/matching_bench.py 128 16 50 0 1 > $O/pmc_c5_write.log 2>&1
rm -f $O/*/*.db
echo done
