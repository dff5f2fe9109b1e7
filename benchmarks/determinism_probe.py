"""Repeats one forward of a ragged batch through the streamed matching, whole and in chunks, and
compares X, iteration counts and permutations bit for bit across repeats.
usage: python benchmarks/determinism_probe.py [repeats=30]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.models import GraphVAE  # noqa: E402

DEV = torch.device("cuda", 0)
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N, B = 24, 12
hb = synthetic_batch("enzymes", B, N, "deg", seed=11)
eps = torch.randn(B, 16, generator=torch.Generator().manual_seed(3))
ref = None
for budget in (None, 3 * _hip.s_block_floats(N, _hip.S_PADDED) * 4):
    torch.manual_seed(4)
    model = GraphVAE(1, 32, 16, N, matching="stream", s_budget_bytes=budget)
    model.precompute_dataset_rho([hb])
    model = model.to(DEV).train()
    batch, e = hb.to(DEV), eps.to(DEV)
    for rep in range(REP):
        with torch.no_grad():
            out = model.forward_with_eps(batch, e)
        torch.cuda.synchronize()
        X = model._engine.debug_tensor("X", B).clone()
        it = model._engine.iters_done.clone()
        perm = model.last_permutations().clone()
        if ref is None:
            ref = (X, it, perm)
            print("R:", hb.r_host, "iters:", it.tolist(), flush=True)
            continue
        dx = (X != ref[0]).flatten(1).any(1).nonzero().flatten().tolist()
        di = (it != ref[1]).nonzero().flatten().tolist()
        dp = (perm != ref[2]).any(1).nonzero().flatten().tolist()
        if dx or di or dp:
            g = (dx or di or dp)[0]
            md = (X[g] - ref[0][g]).abs().max().item()
            print(f"budget={budget} rep={rep}: X differs in graphs {dx}, iters in {di} "
                  f"({[(int(it[k]), int(ref[1][k])) for k in di]}), perm in {dp}; max|dX| of graph {g}: {md:.3e}",
                  flush=True)
print("done")
