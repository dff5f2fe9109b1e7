"""Write rate of the similarity tensor by graph size: the plain producer (gvae_edge_sim) against the
one that also takes the second MPM iteration (inside gvae_match_stream), dense graphs, padded rows.
usage: python benchmarks/producer_sizes.py [sizes...]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 66, 72, 82, 96, 112, 128]
for N in sizes:
    B = max(4, int(2.0e9 / (4.0 * N ** 4)))
    gen = torch.Generator().manual_seed(0)
    up = torch.triu((torch.rand(B, N, N, generator=gen) < 0.5).float(), 1)
    A = (up + up.transpose(1, 2) + torch.eye(N)).to(dev)
    lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
    P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
    Rh = [N] * B
    R = torch.tensor(Rh, dtype=torch.int32, device=dev)
    NS = _hip.node_sim(A, P, R)
    sbytes = 4.0 * B * N ** 2 * N * ((N + 3) & ~3)
    out = []
    for fused in (False, True):
        _hip.profile_enable(True)
        _hip.profile_reset()
        for _ in range(3):
            if fused:
                X = torch.zeros(B, N, N, device=dev)
                _hip.match_stream(A, P, NS, R, Rh, X, 2, 1e-4, False, layout=_hip.S_PADDED)
            else:
                _hip.edge_sim(A, P, NS, R, Rh, _hip.S_PADDED)
        torch.cuda.synchronize()
        ms, n = _hip.profile_read("edge_sim")
        _hip.profile_enable(False)
        out.append(sbytes / (ms / n * 1e-3) / 1e12)
    # one pass over S at the streaming rate (~6.5 TB/s) is what the fused producer saves
    t_plain, t_fused = sbytes / out[0] / 1e12, sbytes / out[1] / 1e12
    print(f"N={N:4d} B={B:5d}: plain {out[0]:5.2f} TB/s, with the iteration {out[1]:5.2f} TB/s; "
          f"extra {1e3 * (t_fused - t_plain):7.3f} ms against {1e3 * sbytes / 6.5e12:7.3f} ms for a pass over S", flush=True)
