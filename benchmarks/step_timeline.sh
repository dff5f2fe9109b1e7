# kernel timeline of one steady-state pipelined step (timed region) of a bench leg
# usage: bash benchmarks/step_timeline.sh [stream|fused] [workload]
set -e
M=${1:-stream}; W=${2:-c2}
O=gpurun_out/tl_${M}_$W; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -- python3 $R/bench.py --workload $W --matching $M --steps ${3:-30} --warmup 5 --no-cpu-baseline --no-fused-leg > $R/$O/bench.json 2> $R/$O/bench.err
cd $R
f=$(ls $O/tr/*/*kernel_trace.csv | head -1)
python3 benchmarks/timeline.py $f ${4:-45} > $O/timeline.txt
rm -rf $O/tr
