"""Latency of the assignment kernel by graph size, on the X a converged MPM produces.
usage: python benchmarks/lsap_sizes.py [B] R1 R2 ...     (dense graphs, one launch of B graphs)
Prints ms per launch (best of 5) and a checksum of the permutations."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402


def main():
    B = int(sys.argv[1])
    dev = torch.device("cuda:0")
    for N in [int(a) for a in sys.argv[2:]]:
        gen = torch.Generator().manual_seed(N)
        up = torch.triu((torch.rand(B, N, N, generator=gen) < 0.5).float(), 1)
        A = (up + up.transpose(1, 2) + torch.eye(N)).to(dev)
        lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
        P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
        Rh = [N] * B
        R = torch.tensor(Rh, dtype=torch.int32, device=dev)
        NS = _hip.node_sim(A, P, R)
        X = _hip.x0_uniform(Rh, N, dev)
        _hip.mpm_fused(A, P, NS, R, N, X, 50, 1e-4, True)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            perm, status = _hip.lsap(X, R, N)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        w = torch.arange(1, N + 1, device=dev, dtype=torch.int64)
        print(f"R={N:4d} B={B}: {best * 1e3:8.3f} ms per launch   status {int(status.sum())}  "
              f"checksum {int((perm * w).sum())}", flush=True)


if __name__ == "__main__":
    main()
