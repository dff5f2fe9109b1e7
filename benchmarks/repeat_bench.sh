# the streamed bench line of a workload several times (run-to-run spread)
# usage: bash benchmarks/repeat_bench.sh <workload> [times=5] [extra bench.py flags]
set -e
W=$1; T=${2:-5}; shift; shift || true
O=gpurun_out/rep; mkdir -p $O
for i in $(seq 1 $T); do
  timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline --no-fused-leg "$@" > $O/${W}_$i.json 2> $O/${W}_$i.err
  python3 -c "import json;d=json.load(open('$O/${W}_$i.json'));print('$W',$i,round(d['ms_per_step'],4),round(d['roofline']['frac'],3))"
done
