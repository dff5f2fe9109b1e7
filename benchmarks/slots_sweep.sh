# pipelined step with 2, 3 and 4 workspaces in flight (GVAE_PIPELINE_SLOTS): fused and streamed c2, c1, c3
set -e
O=gpurun_out/slots; mkdir -p $O
for n in 2 3 4; do
  for w in c2 c1 c3; do
    GVAE_PIPELINE_SLOTS=$n timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline > $O/${w}_$n.json 2> $O/${w}_$n.err
  done
done
python3 - <<'PY'
import json
for n in (2,3,4):
    for w in ("c2","c1","c3"):
        d=json.load(open(f"gpurun_out/slots/{w}_{n}.json"))
        print(n, w, "stream %.3f ms  frac %.3f   fused %.3f ms" % (d["ms_per_step"], d["roofline"]["frac"], d["fused"]["ms_per_step"]))
PY
