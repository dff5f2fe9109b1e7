"""ctypes front-end of the C oracle (oracle/gvae_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> Path:
    so = _HERE / "libgvae_oracle.so"
    src = _HERE / "gvae_oracle.c"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.run(["make", "-C", str(_HERE), "-s", "-B"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(str(build()))
        L.ora_sum_dim1.argtypes = [_f32p, ctypes.c_int, _f32p]
        L.ora_sum_dim1.restype = None
        L.ora_rowsum.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _f32p]
        L.ora_rowsum.restype = None
        L.ora_norm2.argtypes = [_f32p, ctypes.c_int64]
        L.ora_norm2.restype = ctypes.c_float
        L.ora_edge_sim.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int, _f32p,
                                   ctypes.c_void_p]
        L.ora_edge_sim.restype = None
        L.ora_mpm.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                              ctypes.c_int, _f32p]
        L.ora_mpm.restype = ctypes.c_int
        L.ora_mpm_sfree.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_float, ctypes.c_int, _f32p]
        L.ora_mpm_sfree.restype = ctypes.c_int
        L.ora_lsap.argtypes = [_f64p, ctypes.c_int, _i64p]
        L.ora_lsap.restype = ctypes.c_int
        L.ora_assign_perm.argtypes = [_f32p, ctypes.c_int, _i64p]
        L.ora_assign_perm.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def sum_dim1(T: np.ndarray) -> np.ndarray:
    T = _c(T)
    R = T.shape[0]
    assert T.shape == (R, R, R)
    out = np.empty((R, R), np.float32)
    lib().ora_sum_dim1(T, R, out)
    return out


def rowsum(M: np.ndarray) -> np.ndarray:
    M = _c(M)
    R = M.shape[0]
    out = np.empty(R, np.float32)
    lib().ora_rowsum(M, R, M.shape[1], out)
    return out


def norm2(x: np.ndarray) -> np.float32:
    x = _c(x).reshape(-1)
    return np.float32(lib().ora_norm2(x, x.size))


def edge_sim(A: np.ndarray, B: np.ndarray, R: int | None = None) -> np.ndarray:
    """(R,R,R,R) similarity tensor of the leading R x R blocks of A and B."""
    A, B = _c(A), _c(B)
    R = A.shape[0] if R is None else R
    S = np.empty((R, R, R, R), np.float32)
    lib().ora_edge_sim(A, A.shape[1], B, B.shape[1], R, S, None)
    return S


def mpm(S: np.ndarray, X0: np.ndarray, iters: int = 50, tol: float = 1e-4,
        early_stop: bool = True):
    S, X0 = _c(S), _c(X0)
    R = X0.shape[0]
    assert S.shape == (R, R, R, R)
    X = np.empty((R, R), np.float32)
    done = lib().ora_mpm(S, X0, R, iters, np.float32(tol), int(early_stop), X)
    return X, done


def mpm_sfree(A: np.ndarray, B: np.ndarray, X0: np.ndarray | None = None, iters: int = 50,
              tol: float = 1e-4, early_stop: bool = True, R: int | None = None):
    A, B = _c(A), _c(B)
    R = A.shape[0] if R is None else R
    X0 = np.full((R, R), np.float32(1.0 / R), np.float32) if X0 is None else _c(X0)
    X = np.empty((R, R), np.float32)
    done = lib().ora_mpm_sfree(A, A.shape[1], B, B.shape[1], X0, R, iters, np.float32(tol),
                               int(early_stop), X)
    return X, done


def lsap(cost: np.ndarray) -> np.ndarray:
    """col4row of scipy.optimize.linear_sum_assignment(cost) for a square matrix."""
    cost = _c(cost, np.float64)
    n = cost.shape[0]
    col = np.empty(n, np.int64)
    rc = lib().ora_lsap(cost, n, col)
    if rc != 0:
        raise ValueError("cost matrix is infeasible or contains invalid entries")
    return col


def assign_perm(X: np.ndarray) -> np.ndarray:
    """perm with perm[col[i]] = i for the assignment maximising sum X[i, col[i]]."""
    X = _c(X)
    R = X.shape[0]
    perm = np.empty(R, np.int64)
    rc = lib().ora_assign_perm(X, R, perm)
    if rc != 0:
        raise ValueError("matrix contains invalid numeric entries")
    return perm


def x0_uniform(R: int) -> np.ndarray:
    """torch.full((R,R), 1/R): the Python double 1/R rounded to fp32."""
    return np.full((R, R), np.float32(1.0 / R), np.float32)
