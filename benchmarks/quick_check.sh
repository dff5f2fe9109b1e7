# GPU tests of the step, then the bench lines of c2 / c1 / c3 without the CPU baseline
set -e
O=gpurun_out/qc; mkdir -p $O
timeout -k 10 900 python3 -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --workload c1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err
timeout -k 10 300 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python3 - <<'PY'
import json
for k in ("c2","c1","c3"):
    d=json.load(open(f"gpurun_out/qc/bench_{k}.json"))
    print(k, d["value"], d["ms_per_step"], d["roofline"]["frac"], "fused", d.get("fused",{}).get("value"), d.get("fused",{}).get("ms_per_step"))
PY
