#!/usr/bin/env python
"""Live stress of the C oracle against the imported reference (build container only):
random reductions and random matching problems, all compared bit for bit.
usage: python tools/stress_oracle.py [n_matching_cases]"""
import sys
from pathlib import Path

import numpy as np
import scipy.optimize
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import make_golden as mg  # noqa: E402
from oracle import matching as ora  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ref = mg.import_reference()
    m = ref.GraphVAE(1, 4, 2, 4)
    rng = np.random.default_rng(123)
    bad = 0
    for trial in range(3000):
        n = int(rng.integers(1, 300))
        v = (rng.random(n, dtype=np.float32) * rng.choice([0.1, 1.0, 10.0])).astype(np.float32)
        if float(torch.from_numpy(v).norm(p=2)) != float(ora.norm2(v)):
            bad += 1
    print("norm mismatches:", bad)
    bad_r = bad_s = 0
    for trial in range(400):
        R = int(rng.integers(1, 70))
        M = rng.random((R, R), dtype=np.float32)
        bad_r += not np.array_equal(torch.from_numpy(M).sum(1).numpy(), ora.rowsum(M))
        T = rng.random((R, R, R), dtype=np.float32) * (rng.random((R, R, R)) < 0.3)
        bad_s += not np.array_equal(torch.from_numpy(T.astype(np.float32)).sum(dim=1).numpy(),
                                    ora.sum_dim1(T.astype(np.float32)))
    print("rowsum mismatches:", bad_r, " sum(dim=1) mismatches:", bad_s)
    bad_x = bad_p = bad_it = 0
    for trial in range(n_cases):
        R = int(rng.integers(2, 41))
        kind = trial % 4
        if kind == 3:
            import networkx as nx
            a, b = int(rng.integers(1, 6)), int(rng.integers(2, 7))
            g = nx.grid_2d_graph(a, b)
            A = nx.to_numpy_array(g, dtype=np.float32)
            R = A.shape[0]
            np.fill_diagonal(A, 1.0)
        else:
            A = mg.rand_adj(rng, R, [0.1, 0.3, 0.6][kind])
        P = mg.rand_prob(rng, R, float(rng.choice([0.2, 1.0, 3.0])))
        with torch.no_grad():
            S = m._edge_sim_tensor(torch.from_numpy(A), torch.from_numpy(P))
            with mg.IterSpy() as spy:
                X = m._mpm(torch.full((R, R), 1 / R), S)
        Xo, done = ora.mpm_sfree(A, P)
        bad_x += not np.array_equal(Xo, X.numpy())
        bad_it += done != spy.count
        row, col = scipy.optimize.linear_sum_assignment((-X).numpy())
        perm = np.empty_like(col)
        perm[col] = row
        bad_p += not np.array_equal(ora.assign_perm(X.numpy()), perm)
    print(f"matching cases {n_cases}: X mismatches {bad_x}, iteration mismatches {bad_it}, "
          f"perm mismatches {bad_p}")
    return 1 if (bad or bad_r or bad_s or bad_x or bad_p or bad_it) else 0


if __name__ == "__main__":
    sys.exit(main())
