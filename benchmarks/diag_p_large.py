"""Where do the GPU's probabilities sit relative to the reference's and to a float64
recomputation, per graph of the g11 fixtures?  usage: python benchmarks/diag_p_large.py"""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "graph-vae_amd"), str(ROOT / "tests")):
    sys.path.insert(0, p)
from golden_util import load_g11  # noqa: E402
from graphvae.data import collate  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402
from oracle import ref_step as R  # noqa: E402

for N in (64, 126, 128):
    G = load_g11(ROOT / "tests" / "golden" / f"g11_forward_N{N}.npz")
    model = GraphVAE(1, 128, 64, N)
    model.load_state_dict(G.state0, strict=True)
    model.rho_global = G.rho
    model = model.to("cuda:0").train()
    batch = collate(G.graphs, N).to("cuda:0")
    model.forward_with_eps(batch, torch.cat(G.eps, 0).to("cuda:0"))
    P = model._engine.debug_tensor("P", batch.num_graphs).cpu().numpy()
    torch.set_default_dtype(torch.float64)
    st = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in G.state0.items()}
    graphs = [SimpleNamespace(x=g.x.double(), adj_dense=g.adj_dense.double(), num_real_nodes=g.num_real_nodes)
              for g in G.graphs]
    res = R.forward_batch(st, graphs, [e.double() for e in G.eps], train=True, skip_matching=True)
    torch.set_default_dtype(torch.float32)
    for g, Pref in enumerate(G.P):
        r = Pref.shape[0]
        off = ~np.eye(r, dtype=bool)
        T = res.per_graph[g].P[:r, :r].numpy()[off]
        a, b = P[g, :r, :r][off], Pref[off]
        s = np.abs(T).max()
        print(f"N={N} g{g} R={r:3d}: |gpu-ref| {np.abs(a - b).max() / s:.2e}  |gpu-f64| {np.abs(a - T).max() / s:.2e}  "
              f"|ref-f64| {np.abs(b - T).max() / s:.2e}")
