"""Throughput of the sampler's decoder pass (GraphVAE.sample_decode: four GEMMs + the
threshold kernel) with latent vectors resident on the device.
usage: python benchmarks/sample_bench.py [max_nodes] [samples per call] [calls]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae.models import GraphVAE  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GraphVAE(1, 128, 64, N).to(dev).eval()
z = torch.randn(n, 64, device=dev)
model.sample_decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    res = model.sample_decode(z)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
U = N * (N - 1) // 2
# bytes the pass has to move: logits written once and read once, adjacency and masks written
alg = n * (2 * 4 * U + 2 * 4 * N + N * N + N + 4)
flops = 2.0 * n * (64 * 128 * 2 + 128 * U + 128 * N)
print(f"sample_decode max_nodes={N} n={n}: {dt*1e3:.3f} ms per call -> {n/dt:,.0f} graphs/s; "
      f"{alg/dt/1e9:.0f} GB/s algorithmic, {flops/dt/1e12:.2f} TFLOP/s (fp32 MFMA)")
