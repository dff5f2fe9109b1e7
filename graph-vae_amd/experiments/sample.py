#!/usr/bin/env python
"""Sample graphs from a trained GraphVAE on the GPU (mirror of the reference's
experiments/sample.py: same run directory, flags, outputs and printed diagnostics).

  python experiments/sample.py RUN_DIR --num 8 --th_edge 0.5 --th_node 0.5 [--seed 0]

RUN_DIR holds `graphvae_ddp.pt` (state dict with the reference's keys) and `params.yaml`.
The decoder pass of all samples is ONE call into libgvae_hip.so (`GraphVAE.sample_decode`);
the host only turns the returned adjacency bits into graphs: isolated nodes are dropped,
each graph is written to RUN_DIR/samples_fixed/sample_<i>.gml (and .png when matplotlib is
importable).  There is no CPU path for the decode.
"""
from __future__ import annotations

import argparse
import math
import sys
from pathlib import Path

import numpy as np
import torch
import yaml

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from graphvae.eval.metrics import clustering_coef, degree_histogram  # noqa: E402
from graphvae.models.model import GraphVAE  # noqa: E402


def load_model(run_dir: Path, device) -> tuple[GraphVAE, dict]:
    """Model of a run directory; dimensions missing from params.yaml are read off the
    checkpoint's tensor shapes (reference sample.py:20-34)."""
    cfg = yaml.safe_load(open(run_dir / "params.yaml")) or {}
    state = torch.load(run_dir / "graphvae_ddp.pt", map_location="cpu", weights_only=True)
    in_dim = cfg.get("in_dim") or state["conv1.lin.weight"].shape[1]
    pairs = cfg.get("off_diag") or state["dec_edge.2.weight"].size(0)
    max_nodes = cfg.get("max_nodes") or int((1 + math.isqrt(1 + 8 * pairs)) // 2)
    model = GraphVAE(in_dim, cfg["hid_dim"], cfg["z_dim"], max_nodes, pool=cfg.get("pool", "sum"))
    model.load_state_dict(state, strict=True)
    return model.to(device).eval(), cfg


def graphs_from_bits(adj: np.ndarray):
    """(n, N, N) 0/1 -> per sample (kept node ids, edge list over those ids): nodes without an
    edge are dropped, self loops never occur (the diagonal is suppressed on the device)."""
    out = []
    for a in adj:
        a = np.triu(np.maximum(a, a.T), 1)
        rows, cols = np.nonzero(a)
        keep = np.unique(np.concatenate([rows, cols]))
        out.append((keep.tolist(), list(zip(rows.tolist(), cols.tolist()))))
    return out


def write_gml(path: Path, nodes, edges) -> None:
    lines = ["graph ["]
    for v in nodes:
        lines += ["  node [", f"    id {v}", f'    label "{v}"', "  ]"]
    for u, v in edges:
        lines += ["  edge [", f"    source {u}", f"    target {v}", "  ]"]
    lines.append("]")
    path.write_text("\n".join(lines) + "\n")


def write_png(path: Path, nodes, edges) -> None:
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        import networkx as nx
    except ImportError:
        return
    g = nx.Graph()
    g.add_nodes_from(nodes)
    g.add_edges_from(edges)
    plt.figure(figsize=(3, 3))
    nx.draw(g, node_size=60, width=0.8)
    plt.tight_layout()
    plt.savefig(path, dpi=150)
    plt.close()


def main(run_dir: Path, num: int, th_edge: float, th_node: float, device: str, seed: int,
         z: torch.Tensor | None = None):
    if not str(device).startswith("cuda"):
        raise SystemExit("experiments/sample.py decodes on the GPU (the HIP path has no CPU fallback)")
    torch.manual_seed(seed)
    np.random.seed(seed)
    model, cfg = load_model(run_dir, device)
    out_dir = run_dir / "samples_fixed"
    out_dir.mkdir(exist_ok=True)
    if z is None:
        # drawn on the host, as the reference does with --cpu, so that a seed names the same
        # latent vectors on every machine
        z = torch.randn(num, cfg["z_dim"])
    res = model.sample_decode(z.to(device), th_edge, th_node)
    adj = res["adj"].cpu().numpy()
    counts = res["num_nodes"].cpu().tolist()
    graphs = graphs_from_bits(adj)
    N = adj.shape[-1]
    for i, (nodes, edges) in enumerate(graphs):
        r = counts[i]
        a = adj[i].astype(np.float32)
        dummy = np.ones((N, N), dtype=bool)
        dummy[:r, :r] = False
        p_dummy = float(a[dummy].mean()) if dummy.any() else 0.0
        p_real = float(np.triu(a[:r, :r], 1).mean()) if r else float("nan")
        print(f"[{i}] P(edge|dummy)={p_dummy:.3f}  P(edge|real)={p_real:.3f}")
        write_gml(out_dir / f"sample_{i}.gml", nodes, edges)
        write_png(out_dir / f"sample_{i}.png", nodes, edges)
        sub = np.zeros((len(nodes), len(nodes)), dtype=np.float32)
        pos = {v: k for k, v in enumerate(nodes)}
        for u, v in edges:
            sub[pos[u], pos[v]] = sub[pos[v], pos[u]] = 1.0
        hist = degree_histogram(sub)
        print(f"[{i}] nodes={len(nodes):3d} edges={len(edges):3d} "
              f"C={clustering_coef(sub):.3f} max_deg={max(hist) if len(hist) else 0}")
    print(f"{num} graphs saved to {out_dir}")
    return graphs, res


if __name__ == "__main__":
    pa = argparse.ArgumentParser()
    pa.add_argument("run_dir", type=Path)
    pa.add_argument("--num", type=int, default=4)
    pa.add_argument("--th_edge", type=float, default=0.5, help="edge-probability threshold")
    pa.add_argument("--th_node", type=float, default=0.5, help="node-existence threshold")
    pa.add_argument("--cpu", action="store_true", help="accepted for compatibility; refused (GPU only)")
    pa.add_argument("--seed", type=int, default=0)
    args = pa.parse_args()
    dev = "cpu" if args.cpu or not torch.cuda.is_available() else "cuda:0"
    main(args.run_dir, args.num, args.th_edge, args.th_node, dev, args.seed)
