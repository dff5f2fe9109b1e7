# sweep of the tail hand-off threshold of the streamed matching (MiB of live similarity tensors)
for w in ${WORKLOADS:-c2 c3}; do
for mb in ${SIZES:-0 32 64 128 256 512 1024}; do
  echo -n "$w handoff $mb MiB: "
  GVAE_HANDOFF_MB=$mb timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f  streamed iters %.2f of %.2f  tail %.3f ms' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['streamed_iters_per_graph'], r['mean_mpm_iters_per_graph'], r['tail_sfree_ms_per_step']))" || exit 1
done; done
