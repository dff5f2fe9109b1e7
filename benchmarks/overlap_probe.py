"""Where the two streams of an overlapped step are at which time: HIP events on the main
stream (after forward-model, backward, all-reduce, Adam) and on the side stream (after the
matching half), printed as microseconds from the step's start (each step synchronised, so host launch
latency shows; bench.py measures the free-running rate).  GVAE_SIDE_PRIORITY=p
uses a stream of that priority instead of the probed one.  Run with
GVAE_BENCH_FORCE_DIST=1 to include a one-rank RCCL all-reduce.
usage: python benchmarks/overlap_probe.py [stream|fused] [steps=12]"""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "graph-vae_amd"))
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from graphvae.train.train_ddp import TrainStep  # noqa: E402

matching = sys.argv[1] if len(sys.argv) > 1 else "stream"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
force = os.environ.get("GVAE_BENCH_FORCE_DIST") == "1"
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29542")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(42)
model = GraphVAE(1, 128, 64, 32, matching=matching)
host = [synthetic_batch("dense", 256, 32, "deg", seed=b) for b in range(2)]
model.precompute_dataset_rho(host)
model = model.to(dev).train()
batches = [b.to(dev) for b in host]
step = TrainStep(model, 1e-4, 1e-4, 5.0, force_collectives=force)
eng = model._engine_for(batches[0])
if "GVAE_SIDE_PRIORITY" in os.environ:
    side = torch.cuda.Stream(device=dev, priority=int(os.environ["GVAE_SIDE_PRIORITY"]))
else:
    from graphvae.train.train_ddp import concurrent_stream
    side = concurrent_stream(dev)
    print("side stream probe (candidates tried, found):", concurrent_stream.last_probe)
main = torch.cuda.current_stream(dev)
names = ["fwd_model", "backward", "all_reduce", "adam", "match(side)", "joined"]
acc = [0.0] * len(names)
n = 0
for s in range(steps):
    b = batches[s % 2]
    eps = torch.randn(256, 64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    torch.cuda.synchronize()
    ev[0].record(main)
    eng.forward_model(model.flat_parameters(), model.bn_state(), b, eps, model._pos_weight_dev(), True)
    ev[1].record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev[1])
        eng.forward_match_loss()
        ev[5].record(side)
    eng.backward(model.flat_parameters(), model._pos_weight_dev(), step.grads, 1.0, 0.25)
    ev[2].record(main)
    if force:
        dist.all_reduce(step.grads)
    ev[3].record(main)
    eng.adam_step(model.flat_parameters(), step.grads, step.m, step.v, s + 1, 1e-4, clip_max=5.0)
    ev[4].record(main)
    main.wait_event(ev[5])
    ev[6].record(main)
    torch.cuda.synchronize()
    if s >= 2:
        for k in range(len(names)):
            acc[k] += ev[0].elapsed_time(ev[k + 1])
        n += 1
for k, nm in enumerate(names):
    print(f"{nm:12s} done at {1e3 * acc[k] / n:9.1f} us")
