# c3 (ragged, early stop, tail hand-off) and the all-alive ragged throughput by work-group size and
# fine-cut threshold (tuning build)
for cfg in "NT=1024 LOW=256" "NT=256 LOW=256" "NT=256 LOW=512" "NT=256 LOW=1024" "NT=256 LOW=2048" "NT=512 LOW=512"; do
  envs=""; for kv in $cfg; do envs="$envs GVAE_MPM_$kv"; done
  echo -n "== $cfg: "
  env GVAE_HIP_LIB=$PWD/tmp_diag/tune.so $envs timeout -k 10 200 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))" || exit 1
  env GVAE_HIP_LIB=$PWD/tmp_diag/tune.so $envs timeout -k 10 200 python3 benchmarks/ragged_throughput.py 64 12 0 1 | grep seed | sed 's/.*per round; /   /'
done
