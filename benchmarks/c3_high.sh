# c3 and the all-alive ragged throughput by row-grouping threshold (tuning build)
for cfg in "NT=256 LOW=1024 CAP=512 HIGH=4096" "NT=256 LOW=1024 CAP=512 HIGH=2048" "NT=256 LOW=1024 CAP=512 HIGH=1024" "NT=256 LOW=1024 CAP=512 HIGH=512" "NT=256 LOW=512 CAP=512 HIGH=512" "NT=256 LOW=1024 CAP=768 HIGH=1024" "NT=256 LOW=768 CAP=512 HIGH=1024"; do
  envs=""; for kv in $cfg; do envs="$envs GVAE_MPM_$kv"; done
  echo -n "== $cfg: "
  env GVAE_HIP_LIB=$PWD/tmp_diag/tune.so $envs timeout -k 10 200 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))" || exit 1
  env GVAE_HIP_LIB=$PWD/tmp_diag/tune.so $envs timeout -k 10 200 python3 benchmarks/ragged_throughput.py 64 12 0 1 2>/dev/null | grep seed | sed 's/.*per round; /   /'
done
echo "== c2 with NT=256 LOW=1024 CAP=512 (unchanged path expected): "
env GVAE_HIP_LIB=$PWD/tmp_diag/tune.so GVAE_MPM_CAP=512 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))"
