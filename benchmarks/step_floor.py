"""Time of the training step without matching (identity permutation): the cost of everything
around the matching block on the bench's c2 batch.  usage: python benchmarks/step_floor.py"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from graphvae.train.train_ddp import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
for matching in ("none", "fused"):
    torch.manual_seed(42)
    model = GraphVAE(1, 128, 64, 32, matching=matching)
    hb = synthetic_batch("dense", 256, 32, "deg", seed=0)
    model.precompute_dataset_rho([hb])
    model = model.to(dev).train()
    b = hb.to(dev)
    step = TrainStep(model, 1e-4, 1e-4, 5.0)
    for _ in range(5):
        step(b, 0.25, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step(b, 0.25, 1.0)
    torch.cuda.synchronize()
    print(f"matching={matching}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per step")
