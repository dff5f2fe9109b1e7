# sweep of the item-shape thresholds of the streamed MPM on the ENZYMES-shaped workload (tune build)
for cfg in "256 4096" "256 2048" "256 1024" "256 512" "128 1024"; do
  set -- $cfg
  echo "== low $1 high $2"
  GVAE_HIP_LIB=$PWD/tmp_diag/tune.so GVAE_MPM_LOW=$1 GVAE_MPM_HIGH=$2 timeout -k 10 200 python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))"
  GVAE_HIP_LIB=$PWD/tmp_diag/tune.so GVAE_MPM_LOW=$1 GVAE_MPM_HIGH=$2 timeout -k 10 120 python3 benchmarks/matching_bench.py 64 256 50 1 1 | grep "stream wall"
done
