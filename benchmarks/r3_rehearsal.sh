# 2-rank rehearsal (gloo, both ranks on the one GPU) of the pipelined step, 300 steps per matching path, and
# the recorded trainer runs at max_nodes 128 (dense) and 126 (ENZYMES-shaped)
set -e
O=gpurun_out/r3reh; rm -rf $O; mkdir -p $O
for m in fused stream; do
  REH_WATCHDOG=120 timeout -k 10 300 python3 -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 benchmarks/rehearse_dist.py $m pipeline 300 > $O/reh_${m}.log 2>&1
  grep -i "checksum\|identical\|done\|stall" $O/reh_${m}.log | tail -4
done
timeout -k 10 300 python3 benchmarks/trainer_run.py $O/trainer_dense128_b64.json -- --dataset synthetic-dense --feature_type deg --max_nodes 128 --batch_size 64 --epochs 25 --log_dir $O/run_c5 --no_wandb > $O/trainer_dense128_b64.log 2>&1
timeout -k 10 300 python3 benchmarks/trainer_run.py $O/trainer_enzymes126_b64.json -- --dataset synthetic-enzymes --feature_type deg --max_nodes 126 --batch_size 64 --epochs 25 --log_dir $O/run_c3 --no_wandb > $O/trainer_enzymes126_b64.log 2>&1
python3 - <<'PY'
import json
for k in ("dense128_b64", "enzymes126_b64"):
    d = json.load(open(f"gpurun_out/r3reh/trainer_{k}.json"))
    print(k, d["optimisation_steps"], "steps", round(d["wall_s"], 2), "s", round(d["graphs_per_s_incl_startup_validation_and_save"]), "graphs/s", d["peak_device_memory_MiB"], "MiB faults", d["matching_faults"], d["epoch_losses_train_val"][0], d["epoch_losses_train_val"][-1])
PY
