"""Per-iteration latency of the S-free MPM kernel by graph size.
usage: python benchmarks/fused_sizes.py [B] [iters] R1 R2 ...   (GVAE_HIP_LIB=tmp_diag/fstamps.so prints phases)
One launch of B equal-size dense graphs, fixed iteration count, no early stop; with B <= 256
every graph has a CU to itself, so wall time / iters is the latency of one iteration."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402


def main():
    B, iters = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device("cuda:0")
    for N in [int(a) for a in sys.argv[3:]]:
        gen = torch.Generator().manual_seed(N)
        up = torch.triu((torch.rand(B, N, N, generator=gen) < 0.5).float(), 1)
        A = (up + up.transpose(1, 2) + torch.eye(N)).to(dev)
        lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
        P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
        Rh = [N] * B
        R = torch.tensor(Rh, dtype=torch.int32, device=dev)
        NS = _hip.node_sim(A, P, R)
        X0 = _hip.x0_uniform(Rh, N, dev)
        best = 1e9
        for _ in range(3):
            X = X0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _hip.mpm_fused(A, P, NS, R, N, X, iters, 1e-4, False)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"R={N:4d} B={B}: {best * 1e6 / iters:8.2f} us per iteration "
              f"({B * iters * 4.0 * N ** 3 / best / 1e12:6.2f} TFLOP/s)  checksum {float(X.double().sum()):.12f}",
              flush=True)


if __name__ == "__main__":
    main()
