"""Graph statistics of the evaluation scripts (reference graphvae/eval/metrics.py:14-51),
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


def degree_l1(g1, g2, max_deg: int | None = None) -> float:
    """Half the L1 distance between the two degree histograms (total variation); the shorter
    histogram is zero-padded (reference metrics.py:22-28)."""
    h1 = degree_histogram(g1, max_deg)
    h2 = degree_histogram(g2, max_deg or len(h1) - 1)
    m = max(len(h1), len(h2))
    h1 = np.pad(h1, (0, m - len(h1)))
    h2 = np.pad(h2, (0, m - len(h2)))
    return float(np.abs(h1 - h2).sum() / 2.0)


def _hop_distances(a: np.ndarray) -> np.ndarray:
    """All-pairs hop counts by breadth-first frontier expansion; -1 where unreachable."""
    n = a.shape[0]
    adj = a > 0
    dist = np.full((n, n), -1, dtype=np.int64)
    np.fill_diagonal(dist, 0)
    reached = np.eye(n, dtype=bool)
    frontier = reached.copy()
    d = 0
    while frontier.any():
        d += 1
        frontier = ((frontier.astype(np.int64) @ adj.astype(np.int64)) > 0) & ~reached
        dist[frontier] = d
        reached |= frontier
    return dist


def aspl(g) -> float:
    """Average shortest path length; for a disconnected graph the per-component averages
    weighted by component size (reference metrics.py:38-43).  A one-node component counts 0."""
    a = _dense(g)
    n = a.shape[0]
    if n == 0:
        raise ValueError("aspl of a graph without nodes")
    dist = _hop_distances(a)
    label = np.full(n, -1, dtype=np.int64)
    for v in range(n):
        if label[v] < 0:
            label[dist[v] >= 0] = v
    total, weight = 0.0, 0.0
    for c in np.unique(label):
        idx = np.nonzero(label == c)[0]
        k = len(idx)
        mean = dist[np.ix_(idx, idx)].sum() / (k * (k - 1)) if k > 1 else 0.0
        total += k * mean
        weight += k
    return float(total / weight)


def batch_stat(fn, graphs) -> float:
    """mean(fn(g)) over a sequence of graphs (reference metrics.py:49-51)."""
    return float(np.mean([fn(g) for g in graphs]))
