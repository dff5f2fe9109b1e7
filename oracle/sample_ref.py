"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of the decoder loop of the reference sampler, experiments/sample.py:50-73:
dec_edge / dec_node as two-layer perceptrons in fp32 numpy, node existence by threshold on
sigmoid(node logit), edge logits of missing nodes and of the diagonal replaced by -10
(:66-71), threshold on sigmoid (:75), isolated nodes dropped (:79).  Pinned by
tests/golden/g9_sampler.npz, which holds what the imported reference wrote for the same
checkpoint and latent vectors."""
from __future__ import annotations

import numpy as np


def _sigmoid(x: np.ndarray) -> np.ndarray:
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def _mlp(z, w0, b0, w1, b1):
    h = np.maximum(z.astype(np.float32) @ w0.T.astype(np.float32) + b0, 0.0).astype(np.float32)
    return (h @ w1.T.astype(np.float32) + b1).astype(np.float32)


def decode(state: dict, z: np.ndarray, th_edge: float, th_node: float) -> dict:
    """state: reference state-dict arrays.  Returns logits, adjacency bits (n,N,N) uint8,
    node-existence bits (n,N) uint8, node counts and the graphs (kept nodes, edges)."""
    edge = _mlp(z, state["dec_edge.0.weight"], state["dec_edge.0.bias"],
                state["dec_edge.2.weight"], state["dec_edge.2.bias"])          # sample.py:52
    node = _mlp(z, state["dec_node.0.weight"], state["dec_node.0.bias"],
                state["dec_node.2.weight"], state["dec_node.2.bias"])          # sample.py:53
    n, N = node.shape
    iu = np.triu_indices(N, 1)
    exists = _sigmoid(node) > np.float32(th_node)                              # sample.py:58-59
    adj = np.zeros((n, N, N), dtype=np.uint8)
    graphs = []
    for i in range(n):
        lg = np.zeros((N, N), dtype=np.float32)
        lg[iu] = edge[i]
        lg = lg + lg.T                                                         # sample.py:64
        lg[:, ~exists[i]] = -10.0                                              # sample.py:65-66
        lg[~exists[i], :] = -10.0
        np.fill_diagonal(lg, -10.0)                                            # sample.py:69
        bits = _sigmoid(lg) > np.float32(th_edge)                              # sample.py:73
        adj[i] = bits
        up = np.triu(bits | bits.T, 1)
        r, c = np.nonzero(up)
        graphs.append((sorted(set(r.tolist()) | set(c.tolist())), sorted(zip(r.tolist(), c.tolist()))))
    return {"edge_logits": edge, "node_logits": node, "adj": adj, "node_exists": exists.astype(np.uint8),
            "num_nodes": exists.sum(1).astype(np.int32), "graphs": graphs}
