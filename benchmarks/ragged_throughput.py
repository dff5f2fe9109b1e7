"""Streamed MPM on ENZYMES-shaped ragged batches with every graph alive for a fixed number of
iterations (no early stop): the kernel's throughput on ragged items without the chain effects of
convergence.  usage: python benchmarks/ragged_throughput.py [graphs=64] [iters=12] [seed...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402
from graphvae.data import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seeds = [int(a) for a in sys.argv[3:]] or [0, 1, 1000]
N = 126
dev = torch.device("cuda:0")
for seed in seeds:
    batch = synthetic_batch("enzymes", B, N, "deg", seed=seed).to(dev)
    Rh = list(batch.r_host)
    A = batch.adj_dense.contiguous()
    gen = torch.Generator().manual_seed(0)
    lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
    P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
    R = batch.num_real_nodes
    NS = _hip.node_sim(A, P, R)
    S, S_off = _hip.edge_sim(A, P, NS, R, Rh, _hip.S_PADDED)
    ws = torch.empty(_hip.lib().gvae_mpm_stream_ws_bytes(B, N), dtype=torch.uint8, device=dev)
    r = np.array(Rh, dtype=np.float64)
    per_round = float((4 * r ** 4 + 12 * r ** 2).sum())
    for es, label in ((False, "all alive"), (True, "early stop")):
        best, done = 1e9, None
        for _ in range(3):
            X = _hip.x0_uniform(Rh, N, dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = _hip.mpm_stream(S, S_off, R, max(Rh), X, iters if not es else 50, 1e-4, es, ws=ws,
                                   layout=_hip.S_PADDED)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        it = done.cpu().numpy().astype(np.float64)
        alg = float((it * (4 * r ** 4 + 12 * r ** 2)).sum())
        print(f"seed {seed}: max {int(r.max())} nodes, {per_round / 1e6:.0f} MB per round; {label}: "
              f"{int(it.sum())} graph-iterations, {best * 1e3:.3f} ms, {alg / best / 1e12:.2f} TB/s", flush=True)
