for cfg in "TWO=0" "TWO=1" "TWO=1 NT=512" "TWO=0 NT=512" "TWO=1 NT=256" ; do
  envs=""; for kv in $cfg; do envs="$envs GVAE_MPM_$kv"; done
  echo -n "== $cfg: "
  env GVAE_HIP_LIB=$PWD/tmp_diag/tune.so $envs timeout -k 10 200 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))" || exit 1
done
