"""Distance in units in the last place between the GPU's probabilities (prob_kernel: sigmoid of the
decoder logits) and torch's CPU sigmoid of the SAME logits, on the golden cases' logits.
usage: python benchmarks/sigmoid_ulp.py"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "graph-vae_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from golden_util import load_g5, load_g11  # noqa: E402
from graphvae.data import collate  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402


def ulps(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def main():
    gd = ROOT / "tests" / "golden"
    cases = [(f"g5 {c}", load_g5(gd / f"g5_forward_{c}.npz")) for c in ("grid_deg", "grid_id", "rand_deg", "rand_id")]
    cases += [(f"g11 N={n}", load_g11(gd / f"g11_forward_N{n}.npz")) for n in (64, 126, 128)]
    worst = 0
    for label, G in cases:
        d = G.dims
        model = GraphVAE(d["in_dim"], d["H"], d["Z"], d["N"])
        model.load_state_dict(G.state0)
        model.rho_global = G.rho
        model = model.cuda().train()
        batch = collate(G.graphs, d["N"]).to("cuda")
        model.forward_with_eps(batch, torch.cat(G.eps, 0).cuda())
        B, N = batch.num_graphs, d["N"]
        eng = model._engine
        vec = eng.debug_tensor("vec_logits", B).cpu()
        node = eng.debug_tensor("node_logits", B).cpu()
        P = eng.debug_tensor("P", B).cpu().numpy()
        iu = torch.triu_indices(N, N, 1)
        L = torch.zeros(B, N, N)
        L[:, iu[0], iu[1]] = vec
        L = L + L.transpose(1, 2) + torch.diag_embed(node)
        Pc = torch.sigmoid(L).numpy()
        u = ulps(P, Pc)
        worst = max(worst, int(u.max()))
        print(f"{label:12s} {u.size:8d} probabilities: max {int(u.max())} ulp, differing {float((u > 0).mean()) * 100:.2f} %, "
              f"more than 1 ulp {float((u > 1).mean()) * 100:.3f} %")
    print("max over all cases:", worst, "ulp")


if __name__ == "__main__":
    main()
