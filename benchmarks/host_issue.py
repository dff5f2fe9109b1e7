"""Is the fused step bound by the host?  Times K pipelined steps twice: until the loop that issues
them returns (host time) and until the device has finished them.
usage: python benchmarks/host_issue.py [batch=256] [N=32] [steps=50]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from graphvae.train.train_ddp import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = GraphVAE(1, 128, 64, N, matching="fused", mpm_iters=50, mpm_early_stop=True)
host = [synthetic_batch("dense", B, N, "deg", seed=b) for b in range(2)]
model.precompute_dataset_rho(host)
model = model.to(dev).train()
batches = [b.to(dev) for b in host]
for pipeline in (True, False):
    step = TrainStep(model, 1e-4, 1e-4, 5.0, 1, pipeline=pipeline)
    for s in range(10):
        step(batches[s % 2], 0.25, 1.0)
    step.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        step(batches[s % 2], 0.25, 1.0)
    t1 = time.perf_counter()
    step.wait()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} N={N} pipeline={pipeline}: issue {1e3 * (t1 - t0) / K:.3f} ms/step, "
          f"done {1e3 * (t2 - t0) / K:.3f} ms/step", flush=True)
