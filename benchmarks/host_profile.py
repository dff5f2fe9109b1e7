"""cProfile of the host side of the pipelined fused step (where do the ~0.25 ms per step go?).
usage: python benchmarks/host_profile.py [batch=256] [N=32] [steps=200]"""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from graphvae.train.train_ddp import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = GraphVAE(1, 128, 64, N, matching="fused", mpm_iters=50, mpm_early_stop=True)
host = [synthetic_batch("dense", B, N, "deg", seed=b) for b in range(2)]
model.precompute_dataset_rho(host)
model = model.to(dev).train()
batches = [b.to(dev) for b in host]
step = TrainStep(model, 1e-4, 1e-4, 5.0, 1, pipeline=True)
for s in range(10):
    step(batches[s % 2], 0.25, 1.0)
step.wait()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for s in range(K):
    step(batches[s % 2], 0.25, 1.0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
