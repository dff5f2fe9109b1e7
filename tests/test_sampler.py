"""Sampler path (reference experiments/sample.py): the oracle restatement against what the
imported reference wrote (CPU), and the HIP decode against both (GPU).

Tolerance: adjacency and existence bits are exact except where a probability lies within
1e-5 of its threshold (the decoder logits are fp32 GEMM results whose last bits depend on
the summation order); logits 1e-5 relative to the largest entry."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from oracle import sample_ref  # noqa: E402

GOLD = ROOT / "tests" / "golden" / "g9_sampler.npz"


def _load():
    g = np.load(GOLD)
    state = {k[len("state/"):]: g[k] for k in g.files if k.startswith("state/")}
    F, H, Z, N, num, seed = (int(v) for v in g["dims"])
    return g, state, (F, H, Z, N, num, seed)


def _graph_of(g, tag, i):
    return g[f"{tag}/nodes{i}"].tolist(), [tuple(e) for e in g[f"{tag}/edges{i}"].tolist()]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_reproduces_reference_samples(tag):
    g, state, (F, H, Z, N, num, seed) = _load()
    th_e, th_n = g[f"{tag}/th"]
    out = sample_ref.decode(state, g[f"{tag}/z"], th_e, th_n)
    assert np.allclose(out["edge_logits"], g[f"{tag}/edge_logits"], rtol=0, atol=1e-5 * np.abs(g[f"{tag}/edge_logits"]).max())
    assert np.allclose(out["node_logits"], g[f"{tag}/node_logits"], rtol=0, atol=1e-5 * np.abs(g[f"{tag}/node_logits"]).max())
    for i in range(num):
        nodes, edges = _graph_of(g, tag, i)
        assert out["graphs"][i][0] == nodes, (tag, i)
        assert [tuple(e) for e in out["graphs"][i][1]] == edges, (tag, i)


def test_metrics_match_networkx():
    nx = pytest.importorskip("networkx")
    from graphvae.eval.metrics import clustering_coef, degree_histogram
    for seed in range(4):
        gr = nx.erdos_renyi_graph(14, 0.3, seed=seed)
        a = nx.to_numpy_array(gr)
        assert abs(clustering_coef(a) - nx.average_clustering(gr)) < 1e-12
        assert abs(clustering_coef(gr) - nx.average_clustering(gr)) < 1e-12
        ref = np.bincount([d for _, d in gr.degree()]).astype(np.float32)
        assert np.allclose(degree_histogram(a), ref / ref.sum())


def test_path_length_and_degree_distance_match_networkx():
    """aspl / degree_l1 / batch_stat (reference eval/metrics.py:22-28,38-51 restated with
    networkx calls here; sparse graphs so that several components and isolated nodes occur)."""
    nx = pytest.importorskip("networkx")
    from graphvae.eval.metrics import aspl, batch_stat, degree_histogram, degree_l1

    def ref_aspl(g):
        if nx.is_connected(g):
            return nx.average_shortest_path_length(g)
        comps = [g.subgraph(c) for c in nx.connected_components(g)]
        return float(np.average([nx.average_shortest_path_length(c) for c in comps],
                                weights=[len(c) for c in comps]))

    graphs = [nx.erdos_renyi_graph(16, p, seed=s) for s, p in enumerate((0.08, 0.12, 0.2, 0.5))]
    graphs.append(nx.path_graph(7))
    assert any(not nx.is_connected(g) for g in graphs)
    for g in graphs:
        assert abs(aspl(nx.to_numpy_array(g)) - ref_aspl(g)) < 1e-12
        assert abs(aspl(g) - ref_aspl(g)) < 1e-12
    assert abs(batch_stat(aspl, graphs) - np.mean([ref_aspl(g) for g in graphs])) < 1e-12
    for g1, g2 in zip(graphs[:-1], graphs[1:]):
        h1, h2 = degree_histogram(g1), degree_histogram(g2, len(degree_histogram(g1)) - 1)
        m = max(len(h1), len(h2))
        want = np.abs(np.pad(h1, (0, m - len(h1))) - np.pad(h2, (0, m - len(h2)))).sum() / 2.0
        assert abs(degree_l1(g1, g2) - want) < 1e-7
        assert 0.0 <= degree_l1(g1, g2) <= 1.0
    assert degree_l1(graphs[0], graphs[0]) == 0.0


def _near_threshold(logits, th, width=1e-5):
    p = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    return np.abs(p - th) < width


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_decode_matches_reference_samples(tag):
    from graphvae.models import GraphVAE
    g, state, (F, H, Z, N, num, seed) = _load()
    th_e, th_n = (float(v) for v in g[f"{tag}/th"])
    model = GraphVAE(F, H, Z, N)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model = model.to("cuda:0").eval()
    res = model.sample_decode(torch.from_numpy(g[f"{tag}/z"]), th_e, th_n)
    edge, node = res["edge_logits"].cpu().numpy(), res["node_logits"].cpu().numpy()
    assert np.allclose(edge, g[f"{tag}/edge_logits"], rtol=0, atol=1e-5 * np.abs(g[f"{tag}/edge_logits"]).max())
    assert np.allclose(node, g[f"{tag}/node_logits"], rtol=0, atol=1e-5 * np.abs(g[f"{tag}/node_logits"]).max())
    ora = sample_ref.decode(state, g[f"{tag}/z"], th_e, th_n)
    # bits: exact away from the thresholds
    ex, adj = res["node_exists"].cpu().numpy(), res["adj"].cpu().numpy()
    free_n = ~_near_threshold(g[f"{tag}/node_logits"], th_n)
    assert np.array_equal(ex[free_n], ora["node_exists"][free_n])
    if free_n.all():
        assert np.array_equal(res["num_nodes"].cpu().numpy(), ora["num_nodes"])
        iu = np.triu_indices(N, 1)
        for i in range(num):
            lg = np.zeros((N, N))
            lg[iu] = g[f"{tag}/edge_logits"][i]
            free_e = ~_near_threshold(lg + lg.T, th_e)
            assert np.array_equal(adj[i][free_e], ora["adj"][i][free_e]), (tag, i)
            assert np.array_equal(adj[i], adj[i].T) and not adj[i].diagonal().any()
            if free_e.all():          # then the graph itself equals what the reference wrote
                sys.path.insert(0, str(ROOT / "graph-vae_amd" / "experiments"))
                from sample import graphs_from_bits
                nodes, edges = graphs_from_bits(adj[i][None])[0]
                assert (nodes, edges) == _graph_of(g, tag, i), (tag, i)


@pytest.mark.gpu
def test_sampler_cli_on_a_run_directory(tmp_path):
    """The sampler script on a run directory as the trainer leaves it: files written, node
    counts consistent with the returned bits, --cpu refused."""
    import yaml
    sys.path.insert(0, str(ROOT / "graph-vae_amd" / "experiments"))
    import sample as sampler
    g, state, (F, H, Z, N, num, seed) = _load()
    torch.save({k: torch.from_numpy(v) for k, v in state.items()}, tmp_path / "graphvae_ddp.pt")
    yaml.safe_dump({"hid_dim": H, "z_dim": Z, "pool": "sum"}, open(tmp_path / "params.yaml", "w"))
    graphs, res = sampler.main(tmp_path, num, 0.5, 0.5, "cuda:0", seed, z=torch.from_numpy(g["a/z"]))
    assert len(graphs) == num
    for i in range(num):
        assert (tmp_path / "samples_fixed" / f"sample_{i}.gml").exists()
    assert res["adj"].shape == (num, N, N)       # max_nodes recovered from the checkpoint's shapes
    with pytest.raises(SystemExit):
        sampler.main(tmp_path, 1, 0.5, 0.5, "cpu", 0)
