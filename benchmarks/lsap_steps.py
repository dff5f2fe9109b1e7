"""Number of augmenting-path search steps (iterations of the inner `while sink == -1` loop of
SciPy's shortest-augmenting-path solver, one column scanned per step) that the assignment of a
converged MPM result takes, on the inputs of benchmarks/lsap_sizes.py (CPU, the C oracle for the
MPM).  usage: python benchmarks/lsap_steps.py [graphs per size] R1 R2 ...
The kernel executes exactly these steps, one after the other, per graph: time / steps = the
latency of one step."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import matching as ora  # noqa: E402


def count_steps(cost):
    n = cost.shape[0]
    u, v = np.zeros(n), np.zeros(n)
    col4row, row4col, path = -np.ones(n, int), -np.ones(n, int), -np.ones(n, int)
    steps = 0
    for cur in range(n):
        minVal, i, rem = 0.0, cur, list(range(n - 1, -1, -1))
        SR, SC, spc, sink = [], [], np.full(n, np.inf), -1
        while sink == -1:
            steps += 1
            SR.append(i)
            r = minVal + cost[i, rem] - u[i] - v[rem]
            upd = r < spc[rem]
            idx = np.asarray(rem)[upd]
            spc[idx] = r[upd]
            path[idx] = i
            s = spc[rem]
            lowest = s.min()
            cand = np.nonzero(s == lowest)[0]
            free = [k for k in cand if row4col[rem[k]] == -1]
            index = free[-1] if free else cand[0]
            minVal, j = lowest, rem[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC.append(j)
            rem[index] = rem[-1]
            rem.pop()
        u[cur] += minVal
        for r_ in SR:
            if r_ != cur:
                u[r_] += minVal - spc[col4row[r_]]
        for c in SC:
            v[c] -= minVal - spc[c]
        j = sink
        while True:
            r_ = path[j]
            row4col[j] = r_
            col4row[r_], j = j, col4row[r_]
            if r_ == cur:
                break
    return steps, col4row


def main():
    B = int(sys.argv[1])
    for N in [int(a) for a in sys.argv[2:]]:
        gen = torch.Generator().manual_seed(N)
        up = torch.triu((torch.rand(B, N, N, generator=gen) < 0.5).float(), 1)
        A = (up + up.transpose(1, 2) + torch.eye(N)).numpy()
        lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
        P = torch.sigmoid(lg + lg.transpose(1, 2)).numpy()
        tot = []
        for g in range(B):
            X, _ = ora.mpm_sfree(A[g], P[g])
            s, c4r = count_steps(-X.astype(np.float64))
            assert np.array_equal(c4r, ora.lsap(-X.astype(np.float64)))
            tot.append(s)
        print(f"R={N:4d}: steps per graph mean {np.mean(tot):.0f}  max {max(tot)}  (n^2/2 = {N * N // 2})", flush=True)


if __name__ == "__main__":
    main()
