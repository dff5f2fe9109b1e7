# kernel timeline of one steady-state step of the fused leg
set -e
O=gpurun_out/ft; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/bench.py --matching fused --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
f=$(ls $O/tr/*/*kernel_trace.csv | head -1)
python3 benchmarks/timeline.py $f 45 > $O/timeline.txt
rm -rf $O/tr
