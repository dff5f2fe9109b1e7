"""Histogram of MPM iterations per graph on the bench's c2 batch (to model the early-stop tail)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "graph-vae_amd"))
import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
model, batches, host, per_gpu, N, es = bench.make_model_and_batches(args, 0, 1, dev, "stream")
eps = torch.randn(per_gpu, 64, device=dev)
model.forward_with_eps(batches[0], eps)
it = model.last_mpm_iterations().cpu()
h = torch.bincount(it, minlength=51)
print("iterations histogram:", {i: int(c) for i, c in enumerate(h) if c})
alive = [(it >= k).sum().item() for k in range(1, 51)]
print("graphs alive at iteration k=1..50:", alive)
