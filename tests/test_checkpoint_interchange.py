"""Checkpoint interchange with the reference's sampler (experiments/sample.py:20-73) and
trainer (graphvae/train/train_ddp.py:348-351, :412).

The loader below restates what the sampler does with a run directory: read params.yaml,
torch.load the state dict, derive in_dim / max_nodes from tensor shapes when the YAML lacks
them, build GraphVAE(in_dim, hid_dim, z_dim, max_nodes, pool=...), load_state_dict(strict=True),
then decode z ~ N(0, I) through dec_edge / dec_node and vec_to_adj on the chosen device."""
import math
from pathlib import Path

import pytest
import torch
import yaml


def _sampler_load(run_dir: Path, device="cpu"):
    from graphvae.models.model import GraphVAE
    cfg = yaml.safe_load(open(run_dir / "params.yaml"))
    state = torch.load(run_dir / "graphvae_ddp.pt", map_location="cpu", weights_only=True)
    in_dim = cfg.get("in_dim") or state["conv1.lin.weight"].shape[1]
    off_diag = cfg.get("off_diag") or state["dec_edge.2.weight"].size(0)
    max_nodes = cfg.get("max_nodes") or int((1 + math.isqrt(1 + 8 * off_diag)) // 2)
    model = GraphVAE(in_dim, cfg["hid_dim"], cfg["z_dim"], max_nodes, pool=cfg.get("pool", "sum")).to(device)
    model.load_state_dict(state, strict=True)
    model.eval()
    return model, cfg


def _sampler_decode(model, cfg, num, device="cpu", th_edge=0.5, th_node=0.5):
    from graphvae.models.model import vec_to_adj
    with torch.no_grad():
        z = torch.randn(num, cfg["z_dim"], device=device)
        edge_log, node_log = model.dec_edge(z), model.dec_node(z)
    graphs = []
    for i in range(num):
        node_exists = node_log[i].sigmoid() > th_node
        adj_log = vec_to_adj(edge_log[i], model.max_nodes, diag=False)
        adj_log[:, ~node_exists] = -10.0
        adj_log[~node_exists, :] = -10.0
        adj_log.fill_diagonal_(-10.0)
        graphs.append((adj_log.sigmoid() > th_edge).float())
    return edge_log, node_log, graphs


def _reference_shaped_state(F, H, Z, N, seed=0):
    """A state dict with exactly the keys, shapes and dtypes the reference's trainer saves
    (module tree of graphvae/models/model.py:63-94; PyG GCNConv = `bias` + `lin.weight`)."""
    g = torch.Generator().manual_seed(seed)
    U = N * (N - 1) // 2

    def r(*s):
        return torch.randn(*s, generator=g)

    st = {"pos_weight_global": torch.tensor(3.5)}
    for c, fin in (("conv1", F), ("conv2", H)):
        st[f"{c}.bias"], st[f"{c}.lin.weight"] = r(H), r(H, fin)
    for b in ("bn1", "bn2"):
        st[f"{b}.weight"], st[f"{b}.bias"] = r(H), r(H)
        st[f"{b}.running_mean"], st[f"{b}.running_var"] = r(H), r(H).abs() + 0.5
        st[f"{b}.num_batches_tracked"] = torch.tensor(17, dtype=torch.long)
    for name, (o, i) in {"mu": (Z, H), "logvar": (Z, H), "dec_edge.0": (H, Z), "dec_edge.2": (U, H),
                         "dec_node.0": (H, Z), "dec_node.2": (N, H)}.items():
        st[f"{name}.weight"], st[f"{name}.bias"] = r(o, i), r(o)
    return st


def test_reference_checkpoint_loads_and_decodes(tmp_path):
    """A checkpoint in the reference's format (no in_dim / max_nodes in the YAML: the sampler's
    shape fallbacks) loads strict=True; every tensor lands in the flat buffer; the decoder heads
    give what plain matrix arithmetic on the checkpoint's tensors gives."""
    F, H, Z, N = 3, 16, 8, 7
    st = _reference_shaped_state(F, H, Z, N)
    torch.save(st, tmp_path / "graphvae_ddp.pt")
    yaml.safe_dump({"hid_dim": H, "z_dim": Z, "pool": "sum", "dataset": "grid"}, open(tmp_path / "params.yaml", "w"))
    model, cfg = _sampler_load(tmp_path)
    assert (model.in_dim, model.max_nodes) == (F, N)
    back = model.state_dict()
    assert set(back) == set(st)
    for k, v in st.items():
        assert torch.equal(back[k], v), k
        assert back[k].dtype == v.dtype, k
    assert float(model.pos_weight_global) == 3.5
    # the parameters are still views of the one flat buffer after load_state_dict
    flat = model.flat_parameters()
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    for p in model.parameters():
        assert lo <= p.data_ptr() < hi
    torch.manual_seed(3)
    edge_log, node_log, graphs = _sampler_decode(model, cfg, 5)
    torch.manual_seed(3)
    z = torch.randn(5, Z)
    he = torch.relu(z @ st["dec_edge.0.weight"].T + st["dec_edge.0.bias"])
    want = he @ st["dec_edge.2.weight"].T + st["dec_edge.2.bias"]
    assert torch.allclose(edge_log, want, rtol=1e-5, atol=1e-5)
    assert edge_log.shape == (5, N * (N - 1) // 2) and node_log.shape == (5, N)
    for a in graphs:
        assert a.shape == (N, N) and torch.equal(a, a.T) and float(a.diagonal().sum()) == 0.0


def test_own_checkpoint_loads_into_reference_shaped_module(tmp_path):
    """The other direction: what this trainer saves has the keys/shapes a module built the
    reference's way expects (checked against an independent nn.Module with the same tree)."""
    import torch.nn as nn
    from graphvae.models import GraphVAE
    F, H, Z, N = 1, 8, 4, 6

    class _Lin(nn.Module):
        def __init__(s, i, o):
            super().__init__()
            s.weight = nn.Parameter(torch.zeros(o, i))

    class _Conv(nn.Module):
        def __init__(s, i, o):
            super().__init__()
            s.bias = nn.Parameter(torch.zeros(o))
            s.lin = _Lin(i, o)

    class RefTree(nn.Module):
        def __init__(s):
            super().__init__()
            s.conv1, s.bn1 = _Conv(F, H), nn.BatchNorm1d(H)
            s.conv2, s.bn2 = _Conv(H, H), nn.BatchNorm1d(H)
            s.mu, s.logvar = nn.Linear(H, Z), nn.Linear(H, Z)
            s.dec_edge = nn.Sequential(nn.Linear(Z, H), nn.ReLU(), nn.Linear(H, N * (N - 1) // 2))
            s.dec_node = nn.Sequential(nn.Linear(Z, H), nn.ReLU(), nn.Linear(H, N))
            s.register_buffer("pos_weight_global", torch.tensor(1.0))

    m = GraphVAE(F, H, Z, N)
    torch.save({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, tmp_path / "graphvae_ddp.pt")
    ref = RefTree()
    ref.load_state_dict(torch.load(tmp_path / "graphvae_ddp.pt", weights_only=True), strict=True)
    z = torch.randn(3, Z)
    assert torch.equal(ref.dec_edge(z), m.dec_edge(z))


@pytest.mark.gpu
def test_trainer_run_directory_is_loadable_by_the_sampler(tmp_path):
    """Two epochs of the real trainer on the grid data set (GPU), then the sampler's loader and
    decode on the run directory it wrote, on the CPU and on the GPU."""
    from graphvae.train import train_ddp as T
    run = tmp_path / "run"
    T.train({"trainer": {"dataset": "grid", "feature_type": "deg", "epochs": 2, "batch_size": 4,
                         "hid_dim": 16, "z_dim": 8, "log_dir": str(run), "no_wandb": True,
                         "warm_epochs": 1}})
    assert (run / "params.yaml").exists() and (run / "graphvae_ddp.pt").exists()
    for device in ("cpu", "cuda:0"):
        model, cfg = _sampler_load(run, device)
        assert model.max_nodes == 9 and cfg["hid_dim"] == 16
        assert all(torch.isfinite(v).all() for v in model.state_dict().values())
        assert int(model.bn1.num_batches_tracked) > 0
        edge_log, node_log, graphs = _sampler_decode(model, cfg, 4, device)
        assert edge_log.shape == (4, 36) and node_log.shape == (4, 9)
        assert all(torch.equal(a, a.T) for a in graphs)


@pytest.mark.gpu
def test_pipelined_trainer_equals_step_by_step_trainer(tmp_path, capfd):
    """The trainer keeps two steps in flight (matching of one step beside the model half of the
    next); with debug logging it reads the losses after every step and runs them one at a time.
    Same seeds: the checkpoints must be bit-identical and the epoch losses printed the same."""
    import re
    from graphvae.train import train_ddp as T
    outs = []
    for name, debug in (("pipelined", False), ("stepwise", True)):
        run = tmp_path / name
        T.train({"trainer": {"dataset": "synthetic-enzymes", "feature_type": "deg", "epochs": 3,
                             "batch_size": 16, "hid_dim": 16, "z_dim": 8, "max_nodes": 20,
                             "log_dir": str(run), "no_wandb": True, "warm_epochs": 1, "debug": debug}})
        text = capfd.readouterr().out
        losses = re.findall(r"Ep (\d+) \| train ([-\d.naninf]+) \| val ([-\d.naninf]+)", text)
        assert len(losses) == 3, text
        outs.append((losses, torch.load(run / "graphvae_ddp.pt", weights_only=True)))
    assert outs[0][0] == outs[1][0]
    for k, v in outs[0][1].items():
        assert torch.equal(v, outs[1][1][k]), k
