"""Experiment: the optimisation step pipelined across steps.  The matching half of step N
(side stream) overlaps with the model half, backward and Adam of step N+1 (main stream); two
engines alternate so that each step has its own workspace.  Prints free-running ms per step
for the joined schedule (what TrainStep does) and the pipelined one.
usage: python benchmarks/pipeline_probe.py [stream|fused] [steps=40]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "graph-vae_amd"))
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.engine import StepEngine  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from graphvae.train.train_ddp import TrainStep  # noqa: E402

matching = sys.argv[1] if len(sys.argv) > 1 else "stream"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(42)
N, B = 32, 256
model = GraphVAE(1, 128, 64, N, matching=matching)
host = [synthetic_batch("dense", B, N, "deg", seed=b) for b in range(2)]
model.precompute_dataset_rho(host)
model = model.to(dev).train()
batches = [b.to(dev) for b in host]
step = TrainStep(model, 1e-4, 1e-4, 5.0)
engs = [model._engine_for(batches[0]), StepEngine(1, 128, 64, N, dev, matching=matching)]
from graphvae.train.train_ddp import concurrent_stream  # noqa: E402
side = concurrent_stream(dev)
print("side stream probe (candidates tried, found):", concurrent_stream.last_probe)
main = torch.cuda.current_stream(dev)
eps = torch.randn(B, 64, device=dev)


def joined(n):
    for s in range(n):
        step(batches[s % 2], 0.25, 1.0, eps=eps)


ev_model = [torch.cuda.Event(), torch.cuda.Event()]
ev_match = [torch.cuda.Event(), torch.cuda.Event()]
t_adam = [0]


def pipelined(n):
    for s in range(n):
        k = s % 2
        e = engs[k]
        main.wait_event(ev_match[k])          # the matching that last used this slot is done
        e.forward_model(model.flat_parameters(), model.bn_state(), batches[k], eps,
                        model._pos_weight_dev(), True)
        ev_model[k].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_model[k])
            e.forward_match_loss()
            ev_match[k].record(side)
        e.backward(model.flat_parameters(), model._pos_weight_dev(), step.grads, 1.0, 0.25)
        t_adam[0] += 1
        e.adam_step(model.flat_parameters(), step.grads, step.m, step.v, t_adam[0], 1e-4, clip_max=5.0)


step_p = TrainStep(model, 1e-4, 1e-4, 5.0, pipeline=True)


def trainstep_pipelined(n):
    for s in range(n):
        step_p(batches[s % 2], 0.25, 1.0, eps=eps)


def trainstep_pipelined_randn(n):
    for s in range(n):
        step_p(batches[s % 2], 0.25, 1.0)


for name, fn in (("joined", joined), ("pipelined", pipelined), ("TrainStep pipelined", trainstep_pipelined),
                 ("TrainStep pipelined, eps drawn", trainstep_pipelined_randn)):
    for k in range(2):
        ev_match[k].record(main)
    fn(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    print(f"{matching} {name:32s} {1e3 * (time.perf_counter() - t0) / steps:8.3f} ms per step")
