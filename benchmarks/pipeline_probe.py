"""Experiment: free-running ms per step of TrainStep in its three schedules (sequential,
matching half on a second stream and joined, pipelined across steps) on a STATIONARY workload:
learning rate 0, fixed noise, so every step does the same work (real optimisation steps drift:
the MPM needs fewer iterations as training proceeds, which confounds before/after timings).
usage: python benchmarks/pipeline_probe.py [stream|fused] [steps=40] [max_nodes=32] [graphs=256]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "graph-vae_amd"))
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from graphvae.train.train_ddp import TrainStep, concurrent_stream  # noqa: E402

matching = sys.argv[1] if len(sys.argv) > 1 else "stream"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(42)
model = GraphVAE(1, 128, 64, N, matching=matching)
host = [synthetic_batch("dense", B, N, "deg", seed=b) for b in range(2)]
model.precompute_dataset_rho(host)
model = model.to(dev).train()
batches = [b.to(dev) for b in host]
eps = torch.randn(B, 64, device=dev)
for name, kw in (("sequential", dict(overlap_matching=False)), ("two streams, joined", dict()),
                 ("two streams, pipelined", dict(pipeline=True))):
    step = TrainStep(model, 0.0, 0.0, 5.0, **kw)
    for s in range(6):
        step(batches[s % 2], 0.25, 1.0, eps=eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        step(batches[s % 2], 0.25, 1.0, eps=eps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step.wait()
    it = model.last_mpm_iterations().float().mean().item()
    print(f"{matching} N={N} B={B} {name:24s} {1e3 * dt / steps:8.3f} ms per step   (mean MPM iterations {it:.2f}; "
          f"side stream probe {getattr(concurrent_stream, 'last_probe', None)})")
