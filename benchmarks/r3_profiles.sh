# rocprofv3 kernel-trace summaries of the c3 / c4 (50 iterations) / c5 bench lines, both legs of c3
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg > $O/prof_c3.json 2> $O/prof_c3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3f -o c3f -- python3 bench.py --workload c3 --matching fused --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_c3f.json 2> $O/prof_c3f.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 bench.py --workload c4 --mpm-iters 50 --steps 2 --warmup 1 --no-cpu-baseline --no-fused-leg > $O/prof_c4.json 2> $O/prof_c4.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-fused-leg > $O/prof_c5.json 2> $O/prof_c5.err
rm -f $O/*/*.db $O/*/*/*.db
find $O -name "*kernel_trace.csv" -delete
echo done
