"""One launch of the streamed MPM on an ENZYMES-shaped ragged batch (sizes up to 126 nodes,
padded S layout, early stop), for the PMC traffic passes: prints the executed graph-iterations
and the algorithmic bytes sum_g iters_g (4 R_g^4 + 12 R_g^2) of the launch.
usage: python benchmarks/matching_ragged.py [graphs=64] [seed=1000]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402
from graphvae.data import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
N = 126
dev = torch.device("cuda:0")
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
X = _hip.x0_uniform(Rh, N, dev)
done = _hip.mpm_stream(S, S_off, R, max(Rh), X, 50, 1e-4, True, ws=ws, layout=_hip.S_PADDED)
torch.cuda.synchronize()
it = done.cpu().numpy().astype(np.float64)
r = np.array(Rh, dtype=np.float64)
alg = float((it * (4 * r ** 4 + 12 * r ** 2)).sum())
print(f"graphs {B}, sizes max {int(r.max())} mean {r.mean():.1f}; executed graph-iterations {int(it.sum())}; "
      f"algorithmic bytes {alg:.0f}")
