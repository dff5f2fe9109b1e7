"""Diagnostic: GPU intermediates and gradients of a golden case vs float64 recomputation."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "graph-vae_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from golden_util import load_g5, truth64
from graphvae.data import collate
from graphvae.models import GraphVAE

case = sys.argv[1] if len(sys.argv) > 1 else "grid_deg"
G = load_g5(ROOT / "tests" / "golden" / f"g5_forward_{case}.npz")
d = G.dims
model = GraphVAE(d["in_dim"], d["H"], d["Z"], d["N"])
model.load_state_dict(G.state0)
model.rho_global = G.rho
model = model.cuda().train()
batch = collate(G.graphs, d["N"]).to("cuda")
eps = torch.cat(G.eps, 0).cuda()
out = model.forward_with_eps(batch, eps)
out["loss"].backward()
B = batch.num_graphs
eng = model._engine
# float64 intermediates
st = {k: v.double() if v.is_floating_point() else v for k, v in G.state0.items()}
N, H = d["N"], d["H"]
for name in ("Y1", "H1", "Y2", "H2", "gvec"):
    pass
def enc64(g):
    x = g.x.double(); n = x.shape[0]; R = g.num_real_nodes
    A = torch.zeros(n, n, dtype=torch.float64); r = min(R, n)
    A[:r, :r] = (g.adj_dense[:r, :r] != 0).double(); A.fill_diagonal_(1.0)
    dinv = A.sum(1).pow(-0.5); Ah = dinv[:, None] * A * dinv[None, :]
    Y1 = Ah @ (x @ st["conv1.lin.weight"].t())
    pre = Y1 + st["conv1.bias"]; m = pre.mean(0); v = pre.var(0, unbiased=False)
    H1 = torch.relu((pre - m) / torch.sqrt(v + 1e-5) * st["bn1.weight"] + st["bn1.bias"])
    Y2 = Ah @ (H1 @ st["conv2.lin.weight"].t())
    pre2 = Y2 + st["conv2.bias"]; m2 = pre2.mean(0); v2 = pre2.var(0, unbiased=False)
    H2 = torch.relu((pre2 - m2) / torch.sqrt(v2 + 1e-5) * st["bn2.weight"] + st["bn2.bias"])
    return dict(Y1=pre - m, H1=H1, Y2=pre2 - m2, H2=H2, gvec=H2.sum(0) / R, var1=v, var2=v2)   # Y: centred, as the workspace keeps it
for name in ("Y1", "H1", "Y2", "H2", "gvec"):
    t = eng.debug_tensor(name, B).cpu().double()
    worst = 0
    for g, gr in enumerate(G.graphs):
        e = enc64(gr); ref = e[name]
        got = t[g][: ref.shape[0]] if name != "gvec" else t[g]
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    print(f"{name:5s} max rel err vs float64: {worst:.2e}")
for g, gr in enumerate(G.graphs):
    e = enc64(gr)
    print("graph", g, "min var1", float(e["var1"].min()), "min var2", float(e["var2"].min()))
truth = truth64(G)
for n_, p in model.named_parameters():
    t = truth[n_]; m = np.abs(t).max()
    got = p.grad.cpu().numpy().astype(np.float64); gold = G.grads[n_].astype(np.float64)
    print(f"{n_:20s} max|t|={m:.3e} gpu-true={np.abs(got-t).max()/m:.2e} gold-true={np.abs(gold-t).max()/m:.2e} gpu-gold={np.abs(got-gold).max()/m:.2e}")
