"""Diagnostic: where the layer-2 BatchNorm of an ill-conditioned golden case (grid/deg, the 2x2 grid
graph: a column of variance ~ eps) loses accuracy on the GPU, against a float64 recomputation and
against the fp32 CPU restatement.  usage: python benchmarks/diag_bn_column.py [case] [graph]"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "graph-vae_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from golden_util import load_g5
from graphvae.data import collate
from graphvae.models import GraphVAE

case = sys.argv[1] if len(sys.argv) > 1 else "grid_deg"
gi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G = load_g5(ROOT / "tests" / "golden" / f"g5_forward_{case}.npz")
d = G.dims
model = GraphVAE(d["in_dim"], d["H"], d["Z"], d["N"])
model.load_state_dict(G.state0)
model.rho_global = G.rho
model = model.cuda().train()
batch = collate(G.graphs, d["N"]).to("cuda")
eps = torch.cat(G.eps, 0).cuda()
out = model.forward_with_eps(batch, eps)
B = batch.num_graphs
eng = model._engine
gpu = {n: eng.debug_tensor(n, B).cpu().double()[gi] for n in ("Ahat", "Y1", "H1", "Y2", "H2")}


def enc(dt):
    st = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in G.state0.items()}
    g = G.graphs[gi]
    x = g.x.to(dt); n = x.shape[0]; R = int(g.num_real_nodes)
    A = torch.zeros(n, n, dtype=dt); r = min(R, n)
    A[:r, :r] = (g.adj_dense[:r, :r] != 0).to(dt); A.fill_diagonal_(1.0)
    dinv = A.sum(1).pow(-0.5); Ah = dinv[:, None] * A * dinv[None, :]
    def bn(h, nm):
        m = h.mean(0); v = h.var(0, unbiased=False)
        return (h - m) / torch.sqrt(v + 1e-5) * st[nm + ".weight"] + st[nm + ".bias"], m, v
    Y1 = Ah @ (x @ st["conv1.lin.weight"].t()) + st["conv1.bias"]
    h1, _, _ = bn(Y1, "bn1"); H1 = torch.relu(h1)
    T2 = H1 @ st["conv2.lin.weight"].t()
    Y2 = Ah @ T2 + st["conv2.bias"]
    h2, m2, v2 = bn(Y2, "bn2"); H2 = torch.relu(h2)
    return dict(Ahat=Ah, Y1=Y1 - Y1.mean(0), H1=H1, T2=T2, Y2=Y2 - m2, H2=H2, m2=m2, v2=v2, Y2raw=Y2)   # Y: centred, as the workspace keeps it


t, c = enc(torch.float64), enc(torch.float32)
n = t["Y2"].shape[0]
col = int(t["v2"].argmin())
print(f"case {case} graph {gi}: rows {n}, critical column {col}, var {float(t['v2'][col]):.3e}")
for name in ("Ahat", "Y1", "H1", "Y2", "H2"):
    ref = t[name]
    g_ = gpu[name][:n, :ref.shape[1]]
    print(f"{name:5s} max|err|/max|ref|: gpu {float((g_ - ref).abs().max() / ref.abs().max()):.2e}   "
          f"cpu32 {float((c[name].double() - ref).abs().max() / ref.abs().max()):.2e}")
y = t["Y2"][:, col]; spread = float(y.abs().max())
print("critical column of Y2: value", float(t["Y2raw"][:, col].abs().max()), "spread", spread)
print("  row errors / spread, gpu  :", ((gpu["Y2"][:n, col] - y) / spread).numpy().round(6))
print("  row errors / spread, cpu32:", ((c["Y2"][:, col].double() - y) / spread).numpy().round(6))
