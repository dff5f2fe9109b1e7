"""Graph statistics the sampler prints (reference graphvae/eval/metrics.py:14-19,36-37),
computed from a dense 0/1 adjacency with numpy; a networkx graph is accepted as well."""
from __future__ import annotations

import numpy as np


def _dense(g) -> np.ndarray:
    if isinstance(g, np.ndarray):
        a = (g != 0).astype(np.float64)
    else:                                   # networkx graph
        nodes = list(g.nodes())
        index = {v: k for k, v in enumerate(nodes)}
        a = np.zeros((len(nodes), len(nodes)))
        for u, v in g.edges():
            if u != v:
                a[index[u], index[v]] = a[index[v], index[u]] = 1.0
    np.fill_diagonal(a, 0.0)
    return a


def degree_histogram(g, max_deg: int | None = None) -> np.ndarray:
    """Normalised histogram of node degrees over 0..max_deg (default: the largest degree)."""
    deg = _dense(g).sum(1).astype(np.int64)
    top = max_deg or (int(deg.max()) if deg.size else 0)
    hist = np.bincount(deg, minlength=top + 1).astype(np.float32)
    total = hist.sum()
    return hist / total if total else hist


def clustering_coef(g) -> float:
    """Mean local clustering coefficient (nodes of degree < 2 count as 0)."""
    a = _dense(g)
    if a.shape[0] == 0:
        return 0.0
    deg = a.sum(1)
    triangles = np.einsum("ij,jk,ki->i", a, a, a) / 2.0
    pairs = deg * (deg - 1) / 2.0
    local = np.divide(triangles, pairs, out=np.zeros_like(pairs), where=pairs > 0)
    return float(local.mean())
