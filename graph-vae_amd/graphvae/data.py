"""Dense padded graph batches: the input format of the HIP training step.

The reference feeds PyG `Data`/`Batch` objects built by `GraphDataset`
(reference graphvae/train/train_ddp.py:57-112).  PyG is not a dependency here: a
graph is a small record with the same field names (`x`, `edge_index`, `adj_dense`,
`num_real_nodes`, `num_nodes`) and a batch is a `DenseGraphBatch` of stacked padded
tensors, which is what the kernels consume directly.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Iterable, Sequence

import numpy as np
import torch


class GraphData(SimpleNamespace):
    """One graph: x (n_x, F), edge_index (2, E) int64, adj_dense (N, N) with unit diagonal
    on the real nodes, num_real_nodes (0-dim int64 tensor), num_nodes (int)."""

    def to(self, device, non_blocking: bool = False):
        out = GraphData(**self.__dict__)
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                setattr(out, k, v.to(device, non_blocking=non_blocking))
        return out


@dataclass
class DenseGraphBatch:
    """x (B,N,F) zero beyond n_x rows; adj_dense (B,N,N); n_x, num_real_nodes int32 (B,).
    `r_host` keeps the real-node counts on the host so that launches never have to read
    them back from the device."""
    x: torch.Tensor
    adj_dense: torch.Tensor
    n_x: torch.Tensor
    num_real_nodes: torch.Tensor
    r_host: tuple = field(default_factory=tuple)

    @property
    def num_graphs(self) -> int:
        return self.x.shape[0]

    @property
    def max_nodes(self) -> int:
        return self.adj_dense.shape[-1]

    @property
    def device(self):
        return self.x.device

    def to(self, device, non_blocking: bool = False) -> "DenseGraphBatch":
        return DenseGraphBatch(self.x.to(device, non_blocking=non_blocking),
                               self.adj_dense.to(device, non_blocking=non_blocking),
                               self.n_x.to(device, non_blocking=non_blocking),
                               self.num_real_nodes.to(device, non_blocking=non_blocking),
                               self.r_host)

    def pin_memory(self) -> "DenseGraphBatch":
        return DenseGraphBatch(self.x.pin_memory(), self.adj_dense.pin_memory(), self.n_x.pin_memory(),
                               self.num_real_nodes.pin_memory(), self.r_host)

    def index(self, idx) -> "DenseGraphBatch":
        idx_t = torch.as_tensor(idx, dtype=torch.long)
        r = tuple(self.r_host[int(i)] for i in idx_t.tolist())
        dev_idx = idx_t.to(self.x.device)
        return DenseGraphBatch(self.x[dev_idx].contiguous(), self.adj_dense[dev_idx].contiguous(),
                               self.n_x[dev_idx].contiguous(), self.num_real_nodes[dev_idx].contiguous(),
                               r)

    def to_data_list(self) -> list:
        out = []
        for g in range(self.num_graphs):
            nx_ = int(self.n_x[g])
            out.append(GraphData(x=self.x[g, :nx_], adj_dense=self.adj_dense[g],
                                 num_real_nodes=self.num_real_nodes[g].to(torch.long),
                                 num_nodes=int(self.r_host[g])))
        return out


def collate(graphs: Sequence, max_nodes: int | None = None) -> DenseGraphBatch:
    """Stack graph records (GraphData, PyG Data, or anything with x / adj_dense /
    num_real_nodes) into a DenseGraphBatch on the graphs' device."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("cannot collate an empty list of graphs")
    adj0 = graphs[0].adj_dense
    adj0 = adj0.squeeze(0) if adj0.dim() == 3 else adj0
    N = int(max_nodes or adj0.shape[-1])
    F = int(graphs[0].x.shape[-1])
    dev = graphs[0].x.device
    B = len(graphs)
    x = torch.zeros(B, N, F, dtype=torch.float32, device=dev)
    adj = torch.zeros(B, N, N, dtype=torch.float32, device=dev)
    n_x, r = [], []
    for g, d in enumerate(graphs):
        xg = d.x.to(torch.float32)
        a = d.adj_dense
        a = a.squeeze(0) if a.dim() == 3 else a
        if xg.shape[0] > N or a.shape[-1] != N:
            raise ValueError(f"graph {g}: {xg.shape[0]} feature rows / adj {tuple(a.shape)} do not fit max_nodes={N}")
        x[g, : xg.shape[0]] = xg
        adj[g] = a
        n_x.append(int(xg.shape[0]))
        r.append(int(d.num_real_nodes))
    return DenseGraphBatch(x, adj, torch.tensor(n_x, dtype=torch.int32, device=dev),
                           torch.tensor(r, dtype=torch.int32, device=dev), tuple(r))


def as_dense_batch(data, max_nodes: int | None = None) -> DenseGraphBatch:
    """Accepts a DenseGraphBatch, an object with `to_data_list()` (PyG Batch), a list of
    graph records, or a single graph record (the reference's `else [data]` branch,
    graphvae/models/model.py:285)."""
    if isinstance(data, DenseGraphBatch):
        return data
    if hasattr(data, "to_data_list"):
        return collate(data.to_data_list(), max_nodes)
    if isinstance(data, (list, tuple)):
        return collate(data, max_nodes)
    return collate([data], max_nodes)


# --------------------------------------------------------------------------- graphs -> records
def graph_from_adjacency(adj_real: np.ndarray, max_nodes: int, feature_type: str = "deg") -> GraphData:
    """Record for one graph given its (R,R) 0/1 symmetric adjacency without self loops.
    Field semantics follow reference train_ddp.py:67-112."""
    R = int(adj_real.shape[0])
    if R > max_nodes:
        raise ValueError(f"graph with {R} nodes exceeds max_nodes={max_nodes}")
    a = (np.asarray(adj_real) != 0).astype(np.float32)
    a = np.maximum(a, a.T)
    np.fill_diagonal(a, 0.0)
    if feature_type == "id":
        x = torch.eye(max_nodes)[:R].clone()
    elif feature_type == "deg":
        x = torch.zeros(max_nodes, 1)
        x[:R, 0] = torch.from_numpy(a.sum(1))
    else:
        raise ValueError(f"Unknown feature_type: {feature_type}")
    src, dst = np.nonzero(a)
    edge_index = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    adj = torch.zeros(max_nodes, max_nodes)
    adj[:R, :R] = torch.from_numpy(a)
    idx = torch.arange(R)
    adj[idx, idx] = 1.0
    return GraphData(x=x, edge_index=edge_index, adj_dense=adj,
                     num_real_nodes=torch.tensor(R, dtype=torch.long), num_nodes=R)


def graph_from_networkx(g, max_nodes: int, feature_type: str = "deg") -> GraphData:
    import networkx as nx
    nodes = list(g.nodes())
    a = nx.to_numpy_array(g, nodelist=nodes, dtype=np.float32, weight=None)
    return graph_from_adjacency(a, max_nodes, feature_type)


# --------------------------------------------------------------------------- synthetic data
def enzymes_like_sizes(n: int, rng: np.random.Generator, lo: int = 2, hi: int = 126) -> np.ndarray:
    """Node counts shaped like TU-ENZYMES (600 graphs, mean ~32.6 nodes, range 2..126):
    a log-normal fitted to that mean with a long right tail, clipped to [lo, hi]."""
    r = rng.lognormal(mean=np.log(29.0), sigma=0.45, size=n)
    return np.clip(np.rint(r), lo, hi).astype(np.int64)


def synthetic_adjacencies(kind: str, n_graphs: int, max_nodes: int, seed: int = 0) -> list:
    """kind = "dense": R = max_nodes, Erdos-Renyi p = 0.5.
    kind = "enzymes": ENZYMES-shaped sizes, ER with mean degree ~3.8 (>= 1 edge per node).
    kind = "enzymes_filtered": the same with sizes drawn over the full 2..126 range and graphs
    above max_nodes dropped, as the reference's loader drops them (train_ddp.py:147)."""
    rng = np.random.default_rng(seed)
    out = []
    if kind == "dense":
        sizes = np.full(n_graphs, max_nodes)
    elif kind == "enzymes":
        sizes = enzymes_like_sizes(n_graphs, rng, 2, max_nodes)
    elif kind == "enzymes_filtered":
        kept = np.zeros(0, np.int64)
        while kept.size < n_graphs:
            draw = enzymes_like_sizes(2 * n_graphs, rng, 2, 126)
            kept = np.concatenate([kept, draw[draw <= max_nodes]])
        sizes = kept[:n_graphs]
    else:
        raise ValueError("kind must be 'dense', 'enzymes' or 'enzymes_filtered'")
    for R in sizes.tolist():
        p = 0.5 if kind == "dense" else min(1.0, 3.8 / max(R - 1, 1))
        up = np.triu(rng.random((R, R)) < p, 1)
        a = (up | up.T).astype(np.float32)
        # no isolated nodes: connect any such node to its successor (ENZYMES graphs have none)
        for i in np.nonzero(a.sum(1) == 0)[0].tolist():
            j = (i + 1) % R
            if j != i:
                a[i, j] = a[j, i] = 1.0
        out.append(a)
    return out


def synthetic_batch(kind: str, n_graphs: int, max_nodes: int, feature_type: str = "deg",
                    seed: int = 0) -> DenseGraphBatch:
    return collate([graph_from_adjacency(a, max_nodes, feature_type)
                    for a in synthetic_adjacencies(kind, n_graphs, max_nodes, seed)], max_nodes)


class DenseLoader:
    """Mini-batch iterator over a pre-collated dataset kept in pinned host memory.
    `indices_fn(epoch)` returns the sample order of this rank for the epoch."""

    def __init__(self, data: DenseGraphBatch, batch_size: int, indices_fn=None, device=None):
        self.data = data.pin_memory() if (torch.cuda.is_available() and not data.x.is_cuda) else data
        self.batch_size = int(batch_size)
        self.indices_fn = indices_fn or (lambda epoch: list(range(self.data.num_graphs)))
        self.device = device
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        n = len(self.indices_fn(self.epoch))
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterable[DenseGraphBatch]:
        idx = list(self.indices_fn(self.epoch))
        for s in range(0, len(idx), self.batch_size):
            b = self.data.index(idx[s:s + self.batch_size])
            yield b.to(self.device, non_blocking=True) if self.device is not None else b
