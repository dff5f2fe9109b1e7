"""CPU-only tests: the C ABI library loads and exports what include/gvae_hip.h declares,
and the host-side logic (data records, batching, sharding, config handling, module
surface) behaves like the reference's."""
import ctypes
import re
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def test_c_abi_exports_every_declared_symbol():
    from graphvae import _hip
    header = (ROOT / "include" / "gvae_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(gvae_[a-z0-9_]+)\s*\(", header))
    assert {"gvae_edge_sim", "gvae_mpm_stream", "gvae_mpm_fused", "gvae_lsap", "gvae_step_forward",
            "gvae_step_backward", "gvae_adam_step"} <= names
    lib = ctypes.CDLL(str(_hip.lib_path()))
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in gvae_hip.h but not exported"
    lib.gvae_version.restype = ctypes.c_int
    assert lib.gvae_version() >= 100


def test_param_layout_matches_module():
    from graphvae.engine import PARAM_NAMES, param_layout
    from graphvae.models import GraphVAE
    m = GraphVAE(3, 16, 8, 7)
    offs, sizes, tot = param_layout(3, 16, 8, 7)
    byname = dict(m.named_parameters())
    assert [byname[n].numel() for n in PARAM_NAMES] == sizes
    assert all(o % 4 == 0 for o in offs) and tot >= sum(sizes)
    # parameters are views of the flat buffer
    flat = m.flat_parameters()
    for n, o in zip(PARAM_NAMES, offs):
        assert byname[n].data_ptr() == flat[o:].data_ptr()
    byname["mu.bias"].data.fill_(3.0)
    assert float(flat[offs[PARAM_NAMES.index("mu.bias")]]) == 3.0


def test_state_dict_keys_are_the_references():
    """Key set required by the reference's sampler (experiments/sample.py:20-34) and saved by
    its trainer (graphvae/train/train_ddp.py:348-351)."""
    from graphvae.models import GraphVAE
    m = GraphVAE(1, 8, 4, 6)
    keys = set(m.state_dict().keys())
    expect = {"pos_weight_global"}
    for c in ("conv1", "conv2"):
        expect |= {f"{c}.bias", f"{c}.lin.weight"}
    for b in ("bn1", "bn2"):
        expect |= {f"{b}.{k}" for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")}
    for l in ("mu", "logvar", "dec_edge.0", "dec_edge.2", "dec_node.0", "dec_node.2"):
        expect |= {f"{l}.weight", f"{l}.bias"}
    assert keys == expect
    assert m.state_dict()["dec_edge.2.weight"].shape == (15, 8)
    assert m.state_dict()["conv1.lin.weight"].shape == (8, 1)
    # round trip + flat buffer stays the storage
    m2 = GraphVAE(1, 8, 4, 6)
    m2.load_state_dict(m.state_dict(), strict=True)
    assert torch.equal(m2.flat_parameters(), m.flat_parameters())
    assert m2.dec_edge(torch.zeros(2, 4)).shape == (2, 15)      # decoder heads callable on (B, Z)


def test_init_matches_reference_scheme():
    """Xavier-uniform with gain sqrt(2): |w| <= sqrt(2)*sqrt(6/(fan_in+fan_out)); zero biases;
    BatchNorm gamma 1 / beta 0 (reference model.py:123-134)."""
    from graphvae.models import GraphVAE
    m = GraphVAE(5, 32, 16, 10)
    for n, p in m.named_parameters():
        if n.startswith("bn"):
            assert torch.all(p == (1.0 if n.endswith("weight") else 0.0))
        elif p.dim() == 2:
            bound = (2.0 ** 0.5) * (6.0 / (p.shape[0] + p.shape[1])) ** 0.5
            assert p.abs().max() <= bound + 1e-6 and p.abs().max() > 0.5 * bound
        else:
            assert torch.all(p == 0)


def test_vec_to_adj_and_deg_sim():
    from graphvae.models import deg_sim, vec_to_adj
    v = torch.arange(1.0, 7.0)
    a = vec_to_adj(v, 4)
    assert torch.equal(a, torch.tensor([[0, 1, 2, 3], [1, 0, 4, 5], [2, 4, 0, 6], [3, 5, 6, 0.0]]))
    assert torch.equal(vec_to_adj(v, 4, diag=True).diagonal(), torch.ones(4))
    assert vec_to_adj(torch.stack([v, 2 * v]), 4).shape == (2, 4, 4)
    assert torch.equal(deg_sim(torch.tensor([3.0]), torch.tensor([1.0])), torch.tensor([1 / 3.0]))


def test_graph_records_match_reference_dataset_semantics(golden_dir):
    """x / adj_dense / num_real_nodes as the reference's GraphDataset builds them
    (train_ddp.py:67-112); the golden graphs were built by restating that code."""
    import itertools
    import networkx as nx
    from graphvae.train.train_ddp import GraphDataset, build_graph_list
    from golden_util import load_g5
    graphs, max_nodes = build_graph_list("grid", None)
    assert max_nodes == 9 and len(graphs) == 4
    for ft in ("deg", "id"):
        G = load_g5(golden_dir / f"g5_forward_grid_{ft}.npz")
        ds = GraphDataset(graphs, max_nodes, ft)
        assert len(ds) == 4
        for i in range(4):
            rec = ds[i]
            assert torch.equal(rec.x, G.graphs[i].x)
            assert torch.equal(rec.adj_dense, G.graphs[i].adj_dense)
            assert int(rec.num_real_nodes) == G.graphs[i].num_real_nodes == rec.num_nodes
            ei = rec.edge_index
            assert torch.equal(torch.unique(ei, dim=1), torch.unique(G.graphs[i].edge_index, dim=1))
    with pytest.raises(ValueError):
        GraphDataset(graphs, 9, "nope")
    with pytest.raises(ValueError):
        build_graph_list("cora", None)
    # graphs above max_nodes are dropped (train_ddp.py:145-147)
    kept, mn = build_graph_list("grid", 6)
    assert mn == 6 and all(g.number_of_nodes() <= 6 for g in kept) and len(kept) == 3


def test_collate_and_batch_roundtrip():
    from graphvae.data import DenseGraphBatch, as_dense_batch, collate, synthetic_batch
    b = synthetic_batch("enzymes", 7, 30, "id", seed=3)
    assert b.x.shape == (7, 30, 30) and b.adj_dense.shape == (7, 30, 30)
    assert b.n_x.tolist() == list(b.r_host) == b.num_real_nodes.tolist()
    for g, r in enumerate(b.r_host):
        a = b.adj_dense[g]
        assert torch.equal(a, a.T) and torch.all(a.diagonal()[:r] == 1) and torch.all(a[r:] == 0)
        assert torch.equal(b.x[g, :r], torch.eye(30)[:r]) and torch.all(b.x[g, r:] == 0)
    d = synthetic_batch("dense", 3, 8, "deg", seed=1)
    assert d.n_x.tolist() == [8, 8, 8] and d.x.shape == (3, 8, 1)
    assert torch.equal(d.x[..., 0], d.adj_dense.sum(-1) - 1)         # degree without the self loop
    lst = b.to_data_list()
    b2 = as_dense_batch(lst, 30)
    assert torch.equal(b2.x, b.x) and torch.equal(b2.adj_dense, b.adj_dense) and b2.r_host == b.r_host
    one = as_dense_batch(lst[2])
    assert one.num_graphs == 1 and one.r_host == (b.r_host[2],)
    sub = b.index([4, 1])
    assert sub.r_host == (b.r_host[4], b.r_host[1]) and torch.equal(sub.adj_dense[0], b.adj_dense[4])
    with pytest.raises(ValueError):
        collate([])


def test_shard_indices_equals_distributed_sampler():
    from torch.utils.data import DistributedSampler
    from graphvae.train.train_ddp import shard_indices
    data = list(range(23))
    for world in (1, 2, 4, 8):
        for shuffle in (False, True):
            for epoch in (0, 3):
                seen = []
                for rank in range(world):
                    s = DistributedSampler(data, world, rank, shuffle=shuffle)
                    s.set_epoch(epoch)
                    mine = shard_indices(len(data), world, rank, shuffle, epoch)
                    assert mine == list(iter(s)), (world, rank, shuffle, epoch)
                    seen += mine
                assert set(seen) == set(data)


def test_precompute_dataset_rho_matches_reference(golden_dir):
    from graphvae.models import GraphVAE
    from graphvae.train.train_ddp import GraphDataset, build_graph_list
    g = np.load(golden_dir / "g8_rho.npz")
    graphs, n = build_graph_list("grid", None)
    m = GraphVAE(1, 8, 4, n)
    assert m.rho_global is None
    m.precompute_dataset_rho([GraphDataset(graphs, n, "deg").dense()])
    assert abs(m.rho_global - float(g["rho"])) < 1e-12
    assert float(m.pos_weight_global) == float(g["pos_weight"])
    # also accepts per-graph records, as the reference's loader yields them
    m2 = GraphVAE(1, 8, 4, n)
    m2.precompute_dataset_rho([[rec] for rec in GraphDataset(graphs, n, "deg")])
    assert m2.rho_global == m.rho_global


def test_tu_reader(tmp_path):
    from graphvae.train.train_ddp import load_tu_dataset
    raw = tmp_path / "TOY" / "raw"
    raw.mkdir(parents=True)
    (raw / "TOY_graph_indicator.txt").write_text("1\n1\n1\n2\n2\n")
    (raw / "TOY_A.txt").write_text("1, 2\n2, 1\n2, 3\n3, 2\n4, 5\n5, 4\n")
    adjs = load_tu_dataset("TOY", root=str(tmp_path))
    assert [a.shape[0] for a in adjs] == [3, 2]
    assert adjs[0].tolist() == [[0, 1, 0], [1, 0, 1], [0, 1, 0]]
    with pytest.raises(FileNotFoundError):
        load_tu_dataset("ENZYMES", root=str(tmp_path))


def test_cli_yaml_merge_and_defaults(tmp_path, monkeypatch):
    """CLI values override YAML, YAML overrides defaults (reference train_ddp.py:374-384,441-449)."""
    from graphvae.train import train_ddp as T
    seen = {}
    monkeypatch.setattr(T, "_ddp_worker", lambda rank, world, cfg: seen.update(vars(cfg)))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    y = tmp_path / "c.yaml"
    y.write_text("trainer:\n  dataset: grid\n  feature_type: deg\n  epochs: 7\n  lr: 1e-3\n"
                 "  weight_decay: 1e-6\n  warm_epochs: 80\n  debug: true\n")
    T.main(["--config", str(y), "--epochs", "3", "--feature_type", "id", "--log_dir", str(tmp_path / "run"),
            "--no_wandb"])
    assert seen["epochs"] == 3 and seen["feature_type"] == "id" and seen["dataset"] == "grid"
    assert seen["lr"] == 1e-3 and isinstance(seen["lr"], float) and seen["weight_decay"] == 1e-6
    assert seen["warm_epochs"] == 80 and seen["batch_size"] == 64 and seen["hid_dim"] == 128
    assert seen["debug"] is True and seen["use_wandb"] is False
    assert seen["in_dim"] == 9                      # id features on the grid graphs
    assert (tmp_path / "run" / "params.yaml").exists()
    with pytest.raises(ValueError):
        T.train({"trainer": {"dataset": "grid", "lr": "fast", "log_dir": str(tmp_path / "r2")}})


def test_forward_guards_without_gpu():
    from graphvae._hip import GvaeHipError
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    m = GraphVAE(1, 8, 4, 8)
    b = synthetic_batch("dense", 2, 8, "deg")
    with pytest.raises(RuntimeError, match="precompute_dataset_rho"):
        m(b)
    m.rho_global = 0.3
    if not torch.cuda.is_available():
        with pytest.raises(GvaeHipError):
            m(b)
        with pytest.raises(GvaeHipError):
            m._edge_sim_tensor(torch.eye(3), torch.eye(3))


def test_s_chunks_cover_the_batch_within_budget():
    """Chunk boundaries of the streamed matching: consecutive, covering, every chunk within
    the budget, offsets restarting at each chunk; a graph over the budget is an error."""
    from graphvae import _hip
    Rh = [5, 12, 12, 7, 3, 12, 9, 9, 2]
    blk = [_hip.s_block_floats(r, _hip.S_PADDED) for r in Rh]
    assert all(b % 4 == 0 for b in blk) and blk[0] == 5 ** 3 * 8
    budget = 2 * blk[1] + 100
    begin, offs, largest = _hip.s_chunks(Rh, budget)
    assert begin[0] == 0 and begin[-1] == len(Rh) and sorted(set(begin)) == begin
    for c in range(len(begin) - 1):
        tot = 0
        for g in range(begin[c], begin[c + 1]):
            assert offs[g] == tot
            tot += blk[g]
        assert tot <= budget and tot <= largest
    assert _hip.s_chunks(Rh, 10 ** 12)[0] == [0, len(Rh)]
    with pytest.raises(_hip.GvaeHipError):
        _hip.s_chunks(Rh, blk[1] - 4)


def test_priority_guard_rejects_non_default_streams():
    """Streams of a step must all be default-priority streams (train_ddp.require_default_priority):
    None entries are skipped, a stream of another priority class raises before any step runs."""
    from types import SimpleNamespace
    from graphvae import _hip
    from graphvae.train.train_ddp import require_default_priority
    ok = SimpleNamespace(priority=0)
    require_default_priority([ok, None, ok], "test")
    with pytest.raises(_hip.GvaeHipError, match="default priority"):
        require_default_priority([ok, SimpleNamespace(priority=-1)], "test")
