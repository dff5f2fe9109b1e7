set -e
O=gpurun_out/c1r; mkdir -p $O
for i in 1 2 3 4; do
  for n in 2 3; do
    GVAE_PIPELINE_SLOTS=$n timeout -k 10 300 python3 bench.py --workload c1 --no-cpu-baseline --no-fused-leg > $O/c1_${n}_$i.json 2> $O/c1_${n}_$i.err
    python3 -c "import json;d=json.load(open('$O/c1_${n}_$i.json'));print($n,$i,d['ms_per_step'])"
  done
done
