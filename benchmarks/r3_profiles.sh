# measurement set behind profiles/r3_*: bench lines of every single-GPU workload, the c4 iteration sweep,
# rocprofv3 kernel-trace summaries of c2 (both legs), c3 (both legs), c4 (50 iterations) and c5, PMC
# traffic of the streaming kernel (FETCH_SIZE and WRITE_SIZE in separate passes), matrix-core counters of
# the S-free leg's model kernels
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --workload c1 > $O/bench_c1.json 2> $O/bench_c1.err
timeout -k 10 300 python3 bench.py --workload c3 --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err
for it in 50 100 300; do
  timeout -k 10 400 python3 bench.py --workload c4 --mpm-iters $it --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_iters$it.json 2> $O/bench_c4_iters$it.err
done
timeout -k 10 400 python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
echo bench lines done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 bench.py --no-cpu-baseline --no-fused-leg > $O/prof_c2.json 2> $O/prof_c2.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2f -o c2f -- python3 bench.py --matching fused --no-cpu-baseline > $O/prof_c2f.json 2> $O/prof_c2f.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg > $O/prof_c3.json 2> $O/prof_c3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3f -o c3f -- python3 bench.py --workload c3 --matching fused --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_c3f.json 2> $O/prof_c3f.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 bench.py --workload c4 --mpm-iters 50 --steps 2 --warmup 1 --no-cpu-baseline --no-fused-leg > $O/prof_c4.json 2> $O/prof_c4.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-fused-leg > $O/prof_c5.json 2> $O/prof_c5.err
echo kernel traces done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_c2_$c -o p -- python3 benchmarks/matching_bench.py 32 256 50 1 1 > $O/pmc_c2_$c.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_c3_$c -o p -- python3 benchmarks/matching_ragged.py > $O/pmc_c3_$c.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_c5_$c -o p -- python3 benchmarks/matching_bench.py 128 16 50 0 1 > $O/pmc_c5_$c.log 2>&1
done
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c2f -o p -- python3 bench.py --matching fused --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_mfma_c2f.log 2>&1
python3 benchmarks/pmc_summary.py $O/pmc_mfma_c2f $O/pmc_mfma_c2f.csv gvae
rm -f $O/*/*.db $O/*/*/*.db
find $O -name "*kernel_trace.csv" -delete
echo done
