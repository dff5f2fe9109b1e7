# sweep of the largest graph handed from the streamed matching to the S-free kernel (variant
# libraries tmp_diag/h<N>.so built with -DGVAE_HANDOFF_MAX_NODES=<N>) and of the threshold
for w in c3 c1; do
for lib in graph-vae_amd/graphvae/lib/libgvae_hip.so tmp_diag/h44.so tmp_diag/h64.so tmp_diag/h88.so; do
for mb in 128 512 2048; do
  echo -n "$w $lib handoff $mb MiB: "
  GVAE_HIP_LIB=$PWD/$lib GVAE_HANDOFF_MB=$mb timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-fused-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f  mpm %.3f ms  frac %.3f  streamed iters %.2f of %.2f  tail %.3f ms' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['streamed_iters_per_graph'], r['mean_mpm_iters_per_graph'], r['tail_sfree_ms_per_step']))" || exit 1
done; done; done
