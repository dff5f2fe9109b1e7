"""The C oracle (oracle/gvae_oracle.c) against the golden vectors produced from the
imported reference by tools/make_golden.py.  Everything here must be bit-exact."""
import hashlib

import numpy as np
import pytest

from oracle import matching as ora


def _npz(golden_dir, name):
    return np.load(golden_dir / name, allow_pickle=False)


def test_forward_test_known_answer(golden_dir):
    """Reference model.py:447-463: 4-node toy case, col = [2,0,1,3]."""
    g = _npz(golden_dir, "g1_forward_test.npz")
    S = ora.edge_sim(g["A"], g["A1"])
    assert np.array_equal(S, g["S"])
    X, done = ora.mpm(S, ora.x0_uniform(4))
    assert done == int(g["iters_done"])
    assert np.array_equal(X, g["X"])
    col = ora.lsap(-X.astype(np.float64))
    assert col.tolist() == [2, 0, 1, 3] == g["col"].tolist()
    # the value printed by the reference for this case
    np.testing.assert_allclose(X[0], [0.1634, 0.1634, 0.2139, 0.2139], atol=5e-5)


def test_reduction_orders(golden_dir):
    g = _npz(golden_dir, "g7_reduction_orders.npz")
    for R in g["Rs"].tolist():
        T = np.random.default_rng(1000 + R).random((R, R, R), dtype=np.float32)
        M = np.random.default_rng(2000 + R).random((R, R), dtype=np.float32)
        assert np.array_equal(ora.sum_dim1(T), g[f"R{R}/sum_dim1"]), f"sum(dim=1) R={R}"
        assert ora.norm2(M) == g[f"R{R}/norm"], f"norm R={R}"
        assert np.array_equal(ora.rowsum(M), g[f"R{R}/rowsum"]), f"sum(1) R={R}"


def test_norm_all_tail_lengths(golden_dir):
    g = _npz(golden_dir, "g7_reduction_orders.npz")
    lens, vals = g["norm_tail/len"], g["norm_tail/val"]
    assert set((lens % 8).tolist()) == set(range(8))
    for k, (n, ref) in enumerate(zip(lens.tolist(), vals)):
        v = np.random.default_rng(3000 + k).random(n, dtype=np.float32)
        assert ora.norm2(v) == ref, (k, n)


def _names(g):
    return [str(n) for n in g["names"]]


def test_edge_sim_small_full_tensor(golden_dir):
    g = _npz(golden_dir, "g2_matching.npz")
    n = 0
    for name in _names(g):
        if f"{name}/S" in g.files:
            assert np.array_equal(ora.edge_sim(g[f"{name}/A"], g[f"{name}/B"]), g[f"{name}/S"]), name
            n += 1
    assert n >= 10


@pytest.mark.parametrize("big", [False, True])
def test_matching_cases(golden_dir, big):
    """S (by sha256), node block, X, iterations executed and permutation."""
    g = _npz(golden_dir, "g2_matching.npz")
    for name in _names(g):
        A, B = g[f"{name}/A"], g[f"{name}/B"]
        R = A.shape[0]
        if (R > 48) != big:
            continue
        iters, es, done_ref = g[f"{name}/cfg"].tolist()
        tol = float(g[f"{name}/tol"])
        if R <= 65:
            S = ora.edge_sim(A, B)
            assert hashlib.sha256(S.tobytes()).hexdigest() == str(g[f"{name}/S_sha256"]), name
            idx = np.arange(R)
            assert np.array_equal(S[idx[:, None], idx[:, None], idx[None, :], idx[None, :]],
                                  g[f"{name}/node_sim"]), name
            X, done = ora.mpm(S, ora.x0_uniform(R), iters, tol, bool(es))
            assert done == done_ref, name
            assert np.array_equal(X, g[f"{name}/X"]), name
            del S
        X2, done2 = ora.mpm_sfree(A, B, None, iters, tol, bool(es))
        assert done2 == done_ref, name
        assert np.array_equal(X2, g[f"{name}/X"]), name
        assert np.array_equal(ora.assign_perm(X2), g[f"{name}/perm"]), name


def test_lsap_vs_scipy_fixtures(golden_dir):
    g = _npz(golden_dir, "g4_lsap.npz")
    for name in _names(g):
        X = g[f"{name}/X"]
        col = ora.lsap(-X.astype(np.float64))
        assert np.array_equal(col, g[f"{name}/col"]), name


def test_lsap_vs_live_scipy():
    """SciPy is installed at the reference's pin (1.15.3): use it as a live oracle too."""
    scipy_opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(3)
    for trial in range(200):
        R = int(rng.integers(1, 40))
        kind = trial % 4
        if kind == 0:
            X = rng.random((R, R))
        elif kind == 1:
            X = rng.integers(0, 4, (R, R)) / 8.0
        elif kind == 2:
            v = rng.integers(0, 3, (R, R)) / 4.0
            X = (v + v.T) / 2
        else:
            X = np.round(rng.random((R, R)), 1)
        X = X.astype(np.float32)
        _, col = scipy_opt.linear_sum_assignment(-X)
        assert np.array_equal(ora.lsap(-X.astype(np.float64)), col), (trial, R)


def test_sfree_equals_materialised_on_random_cases():
    rng = np.random.default_rng(5)
    for R in (2, 3, 5, 8, 11, 16, 20, 33):
        for _ in range(3):
            up = np.triu((rng.random((R, R)) < 0.4).astype(np.float32), 1)
            A = up + up.T + np.eye(R, dtype=np.float32)
            B = rng.random((R, R)).astype(np.float32)
            B = (B + B.T) / 2
            X1, d1 = ora.mpm(ora.edge_sim(A, B), ora.x0_uniform(R), 50, 1e-4, True)
            X2, d2 = ora.mpm_sfree(A, B, None, 50, 1e-4, True)
            assert d1 == d2 and np.array_equal(X1, X2)
