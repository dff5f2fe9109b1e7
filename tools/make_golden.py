#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the *imported* reference.

Runs only in the build container (needs /root/reference).  The reference's
model module imports torch_geometric at top level (graphvae/models/model.py:33-34),
which is not installed here, so a small in-process stand-in supplies the four
names it imports.  `GCNConv` in the stand-in follows PyG 2.6.1's documented
defaults (add_self_loops=True, normalize=True, bias=True, symmetric
D^-1/2 (A+I) D^-1/2 normalisation, linear without bias -> aggregate -> +bias);
that is third-party arithmetic absent from /root/reference, so every fixture that
touches the encoder is marked "encoder parity unpinned" in its metadata.

Only *data* (inputs and the reference's outputs) is written; no reference source
text leaves /root/reference.

Usage:  python tools/make_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import hashlib
import itertools
import sys
import types
from pathlib import Path

import numpy as np
import scipy.optimize
import torch

REF_ROOT = "/root/reference"


# --------------------------------------------------------------------------- #
# stand-in for torch_geometric (GCNConv / pooling / Batch)                    #
# --------------------------------------------------------------------------- #
class _BareLinear(torch.nn.Module):
    """Weight-only linear layer that is *not* an nn.Linear subclass (PyG's own
    Linear is not one either, which matters for the reference's init walk)."""

    def __init__(self, i: int, o: int) -> None:
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(o, i))
        self.bias = None
        torch.nn.init.xavier_uniform_(self.weight)

    def forward(self, x):
        return x @ self.weight.t()


class _GCNConv(torch.nn.Module):
    def __init__(self, i: int, o: int) -> None:
        super().__init__()
        self.lin = _BareLinear(i, o)
        self.bias = torch.nn.Parameter(torch.zeros(o))

    def forward(self, x, edge_index):
        n = x.size(0)
        src, dst = edge_index[0], edge_index[1]
        keep = src != dst
        src, dst = src[keep], dst[keep]
        loop = torch.arange(n, device=x.device)
        src = torch.cat([src, loop])
        dst = torch.cat([dst, loop])
        w = torch.ones(src.numel(), dtype=x.dtype, device=x.device)
        deg = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, dst, w)
        dinv = deg.pow(-0.5)
        dinv[torch.isinf(dinv)] = 0.0
        norm = dinv[src] * w * dinv[dst]
        h = self.lin(x)
        out = torch.zeros_like(h).index_add_(0, dst, norm.unsqueeze(-1) * h[src])
        return out + self.bias


def _global_add_pool(h, batch):
    nb = int(batch.max()) + 1 if batch.numel() else 1
    return torch.zeros(nb, h.size(1), dtype=h.dtype, device=h.device).index_add_(0, batch, h)


def _global_mean_pool(h, batch):
    nb = int(batch.max()) + 1 if batch.numel() else 1
    cnt = torch.bincount(batch, minlength=nb).clamp(min=1).to(h.dtype).unsqueeze(-1)
    return _global_add_pool(h, batch) / cnt


class _Batch:  # never instantiated; only used in isinstance checks
    pass


def import_reference():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgd = types.ModuleType("torch_geometric.data")
    tgnn.GCNConv = _GCNConv
    tgnn.global_add_pool = _global_add_pool
    tgnn.global_mean_pool = _global_mean_pool
    tgd.Batch = _Batch
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn,
                        "torch_geometric.data": tgd})
    sys.path.insert(0, REF_ROOT)
    import graphvae.models.model as ref_model  # noqa: E402
    return ref_model


# --------------------------------------------------------------------------- #
# input generators (shared with the tests through the stored arrays)          #
# --------------------------------------------------------------------------- #
def rand_adj(rng: np.random.Generator, R: int, p: float) -> np.ndarray:
    up = np.triu((rng.random((R, R)) < p).astype(np.float32), 1)
    A = up + up.T
    np.fill_diagonal(A, 1.0)
    return A


def rand_prob(rng: np.random.Generator, R: int, spread: float = 2.0) -> np.ndarray:
    """Symmetric probabilities in (0,1), like sigmoid of symmetric logits."""
    lg = rng.normal(0.0, spread, (R, R)).astype(np.float32)
    lg = np.triu(lg, 1)
    lg = lg + lg.T + np.diag(rng.normal(0.0, spread, R).astype(np.float32))
    return (1.0 / (1.0 + np.exp(-lg.astype(np.float64)))).astype(np.float32)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class IterSpy:
    """Counts MPM iterations by counting Tensor.amax calls (one per iteration,
    reference model.py:266)."""

    def __enter__(self):
        self.count = 0
        self._orig = torch.Tensor.amax

        def spy(t, *a, **k):
            self.count += 1
            return self._orig(t, *a, **k)

        torch.Tensor.amax = spy
        return self

    def __exit__(self, *exc):
        torch.Tensor.amax = self._orig


# --------------------------------------------------------------------------- #
def gen_forward_test(ref, out: Path):
    """Reference model.py:447-463 known-answer case."""
    m = ref.GraphVAE(in_dim=8, hid_dim=16, z_dim=8, max_nodes=4)
    A = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 0], [0, 1, 1, 1], [0, 0, 1, 1]], dtype=torch.float32)
    A1 = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1]], dtype=torch.float32)
    S = m._edge_sim_tensor(A, A1)
    X0 = torch.full((4, 4), 1 / 4)
    with IterSpy() as spy:
        X = m._mpm(X0, S)
    r, c = scipy.optimize.linear_sum_assignment(-X.numpy())
    np.savez(out / "g1_forward_test.npz", A=A.numpy(), A1=A1.numpy(), S=S.numpy(), X=X.numpy(),
             iters_done=np.int64(spy.count), row=r.astype(np.int64), col=c.astype(np.int64))
    print("G1 forward_test: col =", c, "iters", spy.count)


def matching_cases(rng):
    """(name, A, B, iters, tol, early_stop) tuples."""
    cases = []
    for R in (2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40, 48):
        p = 0.5 if R <= 8 else min(0.5, 3.8 / (R - 1) + 0.1)
        cases.append((f"generic_R{R}", rand_adj(rng, R, p), rand_prob(rng, R), 50, 1e-4, True))
    for R in (6, 16, 32):
        cases.append((f"dense_R{R}_noes", rand_adj(rng, R, 0.5), rand_prob(rng, R, 0.3), 20, 1e-4, False))
        cases.append((f"dense_R{R}_it1", rand_adj(rng, R, 0.5), rand_prob(rng, R, 0.3), 1, 1e-4, True))
    # structured / tie-heavy cases
    for R in (5, 8, 16):
        star = np.eye(R, dtype=np.float32)
        star[0, :] = 1
        star[:, 0] = 1
        cases.append((f"star_R{R}", star, rand_prob(rng, R), 50, 1e-4, True))
        cases.append((f"complete_constB_R{R}", np.ones((R, R), np.float32),
                      np.full((R, R), 0.5, np.float32), 50, 1e-4, True))
        cases.append((f"empty_R{R}", np.eye(R, dtype=np.float32), rand_prob(rng, R), 50, 1e-4, True))
        cases.append((f"generic_constB_R{R}", rand_adj(rng, R, 0.4),
                      np.full((R, R), 0.25, np.float32), 50, 1e-4, True))
    # grid graphs (symmetric: near-ties in X) as in the reference's `grid` dataset
    import networkx as nx
    for a, b in ((2, 3), (3, 3), (2, 2), (4, 5), (5, 6)):
        Ag = nx.to_numpy_array(nx.grid_2d_graph(a, b), dtype=np.float32)
        np.fill_diagonal(Ag, 1.0)
        cases.append((f"grid_{a}x{b}", Ag, rand_prob(rng, a * b, 0.5), 50, 1e-4, True))
    # large: few iterations only (S is R^4 floats in the reference)
    cases.append(("large_R64", rand_adj(rng, 64, 0.1), rand_prob(rng, 64), 50, 1e-4, True))
    cases.append(("large_R65", rand_adj(rng, 65, 0.1), rand_prob(rng, 65), 6, 1e-4, False))
    cases.append(("large_R100", rand_adj(rng, 100, 0.05), rand_prob(rng, 100), 4, 1e-4, False))
    cases.append(("large_R126", rand_adj(rng, 126, 0.04), rand_prob(rng, 126), 3, 1e-4, False))
    cases.append(("large_R128", rand_adj(rng, 128, 0.5), rand_prob(rng, 128, 0.3), 3, 1e-4, False))
    return cases


def gen_matching(ref, out: Path):
    """G2/G3/G4: _edge_sim_tensor, _mpm (+iterations) and SciPy assignment."""
    rng = np.random.default_rng(20250620)
    m = ref.GraphVAE(in_dim=1, hid_dim=4, z_dim=2, max_nodes=4)
    store = {}
    names = []
    for name, A, B, iters, tol, es in matching_cases(rng):
        R = A.shape[0]
        tA, tB = torch.from_numpy(A), torch.from_numpy(B)
        with torch.no_grad():
            S = m._edge_sim_tensor(tA, tB)
            assert S.shape == (R, R, R, R)
            X0 = torch.full((R, R), 1 / R)
            with IterSpy() as spy:
                X = m._mpm(X0, S, iters=iters, tol=tol, early_stop=es)
        Xn = X.numpy()
        row, col = scipy.optimize.linear_sum_assignment((-X).numpy())
        perm = np.empty_like(col)
        perm[col] = row
        idx = np.arange(R)
        node_sim = S[idx[:, None], idx[:, None], idx[None, :], idx[None, :]].numpy()
        names.append(name)
        store[f"{name}/A"] = A
        store[f"{name}/B"] = B
        store[f"{name}/X"] = Xn
        store[f"{name}/node_sim"] = node_sim
        store[f"{name}/col"] = col.astype(np.int64)
        store[f"{name}/perm"] = perm.astype(np.int64)
        store[f"{name}/cfg"] = np.array([iters, int(es), spy.count], dtype=np.int64)
        store[f"{name}/tol"] = np.float64(tol)
        store[f"{name}/S_sha256"] = np.array(sha(S.numpy()))
        store[f"{name}/S_sum"] = np.float64(S.double().sum().item())
        if R <= 8:
            store[f"{name}/S"] = S.numpy()
        print(f"G2-4 {name}: R={R} iters_done={spy.count} perm[:6]={perm[:6]}")
        del S
    store["names"] = np.array(names)
    np.savez_compressed(out / "g2_matching.npz", **store)


def gen_lsap(out: Path):
    """G4 extra: SciPy linear_sum_assignment on tie-heavy / generic fp32 matrices."""
    rng = np.random.default_rng(7)
    store, names = {}, []

    def add(name, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        row, col = scipy.optimize.linear_sum_assignment(-X)
        assert (row == np.arange(X.shape[0])).all()
        names.append(name)
        store[f"{name}/X"] = X
        store[f"{name}/col"] = col.astype(np.int64)

    for R in (1, 2, 3, 4, 5, 8, 13, 16, 31, 32, 64, 100, 128):
        add(f"allequal_R{R}", np.full((R, R), 0.125))
        add(f"rand_R{R}", rng.random((R, R)))
        add(f"smallint_R{R}", rng.integers(0, 3, (R, R)) / 4.0)
        add(f"rowconst_R{R}", np.repeat(rng.random((R, 1)), R, axis=1))
        add(f"colconst_R{R}", np.repeat(rng.random((1, R)), R, axis=0))
        v = rng.random((R, R))
        add(f"sym_R{R}", (v + v.T) / 2)
        blk = np.kron(rng.integers(0, 2, ((R + 1) // 2, (R + 1) // 2)), np.ones((2, 2)))[:R, :R]
        add(f"block_R{R}", blk * 0.5)
    store["names"] = np.array(names)
    np.savez_compressed(out / "g4_lsap.npz", **store)
    print("G4 lsap:", len(names), "cases")


def gen_reduction_orders(out: Path):
    """G7: the three CPU reduction orders the matching result depends on."""
    rng = np.random.default_rng(11)
    store = {}
    Rs = (2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40, 47, 48, 63, 64, 65, 100, 126, 128)
    for R in Rs:
        # inputs are not stored: tests rebuild them from default_rng(1000+R) / (2000+R)
        Tg = np.random.default_rng(1000 + R).random((R, R, R), dtype=np.float32)
        store[f"R{R}/sum_dim1"] = torch.from_numpy(Tg).sum(dim=1).numpy()
        Mg = np.random.default_rng(2000 + R).random((R, R), dtype=np.float32)
        store[f"R{R}/norm"] = np.float32(torch.from_numpy(Mg).norm(p=2).item())
        store[f"R{R}/rowsum"] = torch.from_numpy(Mg).sum(1).numpy()
    store["Rs"] = np.array(Rs, dtype=np.int64)
    # norm with every tail length n % 8 (inputs rebuilt in the tests from default_rng(3000+k))
    lens, vals = [], []
    for k in range(240):
        n = 8 * (k % 5) + (k % 8) + (8 if k % 5 == 0 and k % 8 == 0 else 0)
        v = np.random.default_rng(3000 + k).random(n, dtype=np.float32)
        lens.append(n)
        vals.append(np.float32(torch.from_numpy(v).norm(p=2).item()))
    store["norm_tail/len"] = np.array(lens, dtype=np.int64)
    store["norm_tail/val"] = np.array(vals, dtype=np.float32)
    np.savez_compressed(out / "g7_reduction_orders.npz", **store)
    print("G7 reduction orders:", len(Rs), "sizes")


# --------------------------------------------------------------------------- #
def nx_to_graph(g, max_nodes: int, feature_type: str):
    """Restates the data contract of reference train_ddp.py:67-112 with numpy/torch
    only (that module cannot be imported here: it needs tensorboard and PyG)."""
    import networkx as nx
    mapping = {n: i for i, n in enumerate(g.nodes())}
    g = nx.relabel_nodes(g, mapping)
    R = g.number_of_nodes()
    if feature_type == "id":
        x = torch.eye(max_nodes)[:R]
    else:
        x = torch.zeros(max_nodes, 1)
        x[:R, 0] = torch.tensor([d for _, d in g.degree()], dtype=torch.float32)
    e = [(u, v) for u, v in g.edges()] + [(v, u) for u, v in g.edges()]
    e = sorted(set(e))
    ei = torch.tensor(e, dtype=torch.long).t().contiguous() if e else torch.zeros(2, 0, dtype=torch.long)
    adj = torch.zeros(max_nodes, max_nodes)
    if e:
        adj[ei[0], ei[1]] = 1
    adj = torch.maximum(adj, adj.T)
    adj[torch.arange(R), torch.arange(R)] = 1.0
    return types.SimpleNamespace(x=x, edge_index=ei, adj_dense=adj,
                                 num_real_nodes=torch.tensor(R, dtype=torch.long))


def graphs_for_forward(kind: str):
    import networkx as nx
    if kind == "grid":
        return [nx.grid_2d_graph(i, j) for i, j in itertools.product(range(2, 4), repeat=2)], 9
    rs = np.random.RandomState(5)
    gs = []
    for R in (5, 12, 9, 12, 7, 3):
        while True:
            g = nx.gnp_random_graph(R, 0.4, seed=int(rs.randint(1 << 30)))
            if min(dict(g.degree()).values()) > 0:
                break
        gs.append(g)
    return gs, 12


def run_reference_step(ref, model, graphs, eps_list, train: bool, method: str = "forward"):
    """One reference forward (`forward`, or the legacy `forward_old`, model.py:378-440)
    over a list of single-graph inputs, reproducing the
    batch reduction of model.py:368-373 (a `Batch` cannot be built without PyG,
    so graphs are fed one at a time through the `else [data]` branch and the
    per-graph losses are averaged exactly as forward does for a batch)."""
    perms = []
    captured = {"P": [], "X": []}
    orig_es, orig_mpm = model._edge_sim_tensor, model._mpm

    def es_spy(A, B):
        captured["P"].append(B.detach().clone().numpy())
        return orig_es(A, B)

    def mpm_spy(X0, S, *a, **k):
        X = orig_mpm(X0, S, *a, **k)
        captured["X"].append(X.detach().clone().numpy())
        return X

    model._edge_sim_tensor, model._mpm = es_spy, mpm_spy
    orig_lsa = scipy.optimize.linear_sum_assignment
    orig_randn_like = torch.randn_like
    it = iter(eps_list)

    def lsa_spy(cost, *a, **k):
        r, c = orig_lsa(cost, *a, **k)
        p = np.empty_like(c)
        p[c] = r
        perms.append(p.astype(np.int64))
        return r, c

    def fixed_randn_like(t, *a, **k):
        e = next(it)
        assert e.shape == t.shape
        return e.clone()

    scipy.optimize.linear_sum_assignment = lsa_spy
    torch.randn_like = fixed_randn_like
    try:
        model.train(train)
        outs = [model(g) if method == "forward" else getattr(model, method)(g) for g in graphs]
    finally:
        scipy.optimize.linear_sum_assignment = orig_lsa
        torch.randn_like = orig_randn_like
        del model._edge_sim_tensor, model._mpm
    rec = torch.stack([o["rec"] for o in outs]).mean()
    kl = torch.stack([o["kl"] for o in outs]).mean()
    max_logit = torch.stack([o["max_logit"] for o in outs]).max()
    return {"rec": rec, "kl": kl, "loss": rec + kl, "max_logit": max_logit,
            "logvar_max": outs[-1]["logvar_max"],
            "rec_i": torch.stack([o["rec"] for o in outs]).detach(),
            "kl_i": torch.stack([o["kl"] for o in outs]).detach(),
            "P": captured["P"], "X": captured["X"]}, perms


def gen_forward(ref, out: Path):
    """G5: full forward + backward; G6: three optimizer steps with the schedule of
    reference train_ddp.py:282-292,316."""
    for kind, ftype in (("grid", "deg"), ("grid", "id"), ("rand", "deg"), ("rand", "id")):
        nxg, N = graphs_for_forward(kind)
        graphs = [nx_to_graph(g, N, ftype) for g in nxg]
        in_dim = graphs[0].x.size(-1)
        H, Z = 16, 8
        torch.manual_seed(42)
        model = ref.GraphVAE(in_dim, H, Z, N)
        # perturb BN affine + biases so that every gradient path is exercised
        gen = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if n_.endswith("bias") or n_.startswith("bn"):
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
        pos_e = sum(float(np.triu(g.adj_dense.numpy()[:int(g.num_real_nodes), :int(g.num_real_nodes)], 1).sum()) for g in graphs)
        pairs = sum(int(g.num_real_nodes) * (int(g.num_real_nodes) - 1) / 2 for g in graphs)
        rho = pos_e / (pairs + 1e-8)
        model.rho_global = rho
        model.pos_weight_global.fill_(float(np.clip((1 - rho) / (rho + 1e-8), 1.0, 20.0)))
        state0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        eps = [torch.randn(1, Z, generator=gen) for _ in graphs]

        res, perms = run_reference_step(ref, model, graphs, eps, train=True)
        res["loss"].backward()
        store = {"meta": np.array(f"{kind}/{ftype}; encoder parity unpinned (PyG GCNConv restated)"),
                 "N": np.int64(N), "H": np.int64(H), "Z": np.int64(Z), "in_dim": np.int64(in_dim),
                 "rho": np.float64(rho)}
        for i, g in enumerate(graphs):
            store[f"g{i}/x"] = g.x.numpy()
            store[f"g{i}/edge_index"] = g.edge_index.numpy()
            store[f"g{i}/adj_dense"] = g.adj_dense.numpy()
            store[f"g{i}/R"] = np.int64(int(g.num_real_nodes))
            store[f"g{i}/eps"] = eps[i].numpy()
            store[f"g{i}/perm"] = perms[i]
            store[f"g{i}/P"] = res["P"][i]      # sigmoid(logits)[:R,:R] handed to the matching
            store[f"g{i}/X"] = res["X"][i]      # its MPM result
        store["n_graphs"] = np.int64(len(graphs))
        for k, v in state0.items():
            store[f"state0/{k}"] = v
        for k in ("rec", "kl", "loss", "max_logit", "logvar_max", "rec_i", "kl_i"):
            store[f"out/{k}"] = res[k].detach().numpy()
        for n_, p in model.named_parameters():
            store[f"grad/{n_}"] = p.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running" in k or "num_batches" in k:
                store[f"state1/{k}"] = v.numpy().copy()

        # eval-mode forward (running stats), same eps
        with torch.no_grad():
            res_e, perms_e = run_reference_step(ref, model, graphs, eps, train=False)
        for k in ("rec", "kl", "loss"):
            store[f"out_eval/{k}"] = res_e[k].numpy()
        for i, p in enumerate(perms_e):
            store[f"g{i}/perm_eval"] = p

        # G6: three optimizer steps from state0 (fresh model), schedule restated
        torch.manual_seed(42)
        model2 = ref.GraphVAE(in_dim, H, Z, N)
        model2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state0.items()})
        model2.rho_global = rho
        lr, wd, beta0, beta_final, clip_max, anneal_steps = 1e-2, 1e-4, 0.0, 0.5, 5.0, 2
        opt = torch.optim.Adam(model2.parameters(), lr=lr, weight_decay=wd)
        losses = []
        for gs in range(3):
            eps_s = [torch.randn(1, Z, generator=gen) for _ in graphs]
            for i, e in enumerate(eps_s):
                store[f"opt/eps{gs}_{i}"] = e.numpy()
            res_s, _ = run_reference_step(ref, model2, graphs, eps_s, train=True)
            frac = min(1.0, gs / anneal_steps)
            beta = beta0 + (beta_final - beta0) * frac
            loss = res_s["rec"] + beta * res_s["kl"]
            for pg in opt.param_groups:
                pg["lr"] = lr * frac
            opt.zero_grad(set_to_none=True)
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(model2.parameters(), clip_max)
            opt.step()
            losses.append([loss.item(), res_s["rec"].item(), res_s["kl"].item(), gn.item()])
        store["opt/hparams"] = np.array([lr, wd, beta0, beta_final, clip_max, anneal_steps], dtype=np.float64)
        store["opt/losses"] = np.array(losses, dtype=np.float64)
        for k, v in model2.state_dict().items():
            store[f"opt/state3/{k}"] = v.detach().numpy().copy()
        np.savez_compressed(out / f"g5_forward_{kind}_{ftype}.npz", **store)
        print(f"G5/G6 {kind}/{ftype}: loss={res['loss'].item():.6f} rec={res['rec'].item():.6f} "
              f"kl={res['kl'].item():.6f} opt losses={[round(l[0], 5) for l in losses]}")


def gen_forward_old(ref, out: Path):
    """G10: the legacy loss `forward_old` (reference model.py:378-440) on the rand/deg graphs
    of G5, from the same initial state and the same noise: losses, permutations and all
    parameter gradients.  Inputs are not stored again; the fixture records checksums of the
    G5 inputs it was generated from."""
    kind, ftype = "rand", "deg"
    nxg, N = graphs_for_forward(kind)
    graphs = [nx_to_graph(g, N, ftype) for g in nxg]
    H, Z = 16, 8
    base = np.load(out / f"g5_forward_{kind}_{ftype}.npz")
    torch.manual_seed(42)
    model = ref.GraphVAE(1, H, Z, N)
    model.load_state_dict({k[len("state0/"):]: torch.from_numpy(base[k].copy())
                           for k in base.files if k.startswith("state0/")})
    model.rho_global = float(base["rho"])
    eps = [torch.from_numpy(base[f"g{i}/eps"].copy()) for i in range(len(graphs))]
    for i, g in enumerate(graphs):
        assert np.array_equal(base[f"g{i}/adj_dense"], g.adj_dense.numpy())
    res, perms = run_reference_step(ref, model, graphs, eps, train=True, method="forward_old")
    (res["rec"] + res["kl"]).backward()
    store = {"meta": np.array("forward_old on the inputs, initial state and noise of g5_forward_rand_deg"),
             "inputs_sha": np.array(sha(np.concatenate([base[f"g{i}/adj_dense"].ravel()
                                                        for i in range(len(graphs))])))}
    for k in ("rec", "kl", "max_logit", "logvar_max", "rec_i", "kl_i"):
        store[f"out/{k}"] = res[k].detach().numpy()
    for i, p in enumerate(perms):
        store[f"g{i}/perm"] = p
    for n_, p in model.named_parameters():
        store[f"grad/{n_}"] = p.grad.numpy().copy()
    np.savez_compressed(out / "g10_forward_old_rand_deg.npz", **store)
    d = float(res["rec"]) - float(base["out/rec"])
    print(f"G10 forward_old rand/deg: rec={res['rec'].item():.6f} (forward: {float(base['out/rec']):.6f}, "
          f"diff {d:+.3e}) kl={res['kl'].item():.6f}")


def gen_individual_rho(out: Path):
    """G12: `forward_individual_rho` of the reference's graphvae/models/model_legacy.py:257-363 on
    the rand/id graphs, initial state and noise of G5: the variant that permutes the ground truth
    only (the prediction stays in place), weights positives by each graph's own density and runs
    the matching over all max_nodes rows.  The legacy class has one decoder `dec` where the
    current one has `dec_edge` (same shape) and `dec_node` (unused here); `id` features, for which
    its mean pooling equals the current sum / R pooling."""
    import importlib
    legacy = importlib.import_module("graphvae.models.model_legacy")
    kind, ftype = "rand", "id"
    nxg, N = graphs_for_forward(kind)
    graphs = [nx_to_graph(g, N, ftype) for g in nxg]
    for g in graphs:
        g.num_nodes = int(g.num_real_nodes)
    H, Z = 16, 8
    base = np.load(out / f"g5_forward_{kind}_{ftype}.npz")
    torch.manual_seed(42)
    model = legacy.GraphVAE(N, H, Z, N)
    sd = {}
    for k in base.files:
        if not k.startswith("state0/"):
            continue
        name = k[len("state0/"):]
        if name.startswith("dec_node") or name == "pos_weight_global":
            continue
        sd[name.replace("dec_edge", "dec")] = torch.from_numpy(base[k].copy())
    model.load_state_dict(sd)
    eps = [torch.from_numpy(base[f"g{i}/eps"].copy()) for i in range(len(graphs))]
    for i, g in enumerate(graphs):
        assert np.array_equal(base[f"g{i}/adj_dense"], g.adj_dense.numpy())
    res, perms = run_reference_step(legacy, model, graphs, eps, train=True, method="forward_individual_rho")
    (res["rec"] + res["kl"]).backward()
    store = {"meta": np.array("model_legacy.forward_individual_rho on the inputs, initial state and noise of "
                              "g5_forward_rand_id (dec <- dec_edge)"),
             "inputs_sha": np.array(sha(np.concatenate([base[f"g{i}/adj_dense"].ravel()
                                                        for i in range(len(graphs))])))}
    for k in ("rec", "kl", "max_logit", "logvar_max", "rec_i", "kl_i"):
        store[f"out/{k}"] = res[k].detach().numpy()
    for i, p in enumerate(perms):
        store[f"g{i}/perm"] = p
        store[f"g{i}/P"] = res["P"][i]
        store[f"g{i}/X"] = res["X"][i]
    for n_, p in model.named_parameters():
        store[f"grad/{n_.replace('dec.', 'dec_edge.')}"] = p.grad.numpy().copy()
    np.savez_compressed(out / "g12_individual_rho_rand_id.npz", **store)
    print(f"G12 forward_individual_rho rand/id: rec={res['rec'].item():.6f} kl={res['kl'].item():.6f} "
          f"(forward: rec {float(base['out/rec']):.6f}); perms {[p.tolist() for p in perms[:2]]}")


def gen_forward_large(ref, out: Path):
    """G11: forward + backward of the imported reference at the max_nodes of the larger BASELINE
    configurations (64, 126, 128) with the trainer's default widths (hid 128, z 64), on
    ENZYMES-shaped graphs of at most 48 real nodes (what the reference's CPU path can hold: it
    keeps every MPM iteration's R^4 product for autograd).  Weights come from
    tests/golden_util.seeded_state (a function of the stored seed); gradients of padded pairs
    are exact zeros, so the large tensors compress."""
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
    from golden_util import graph_from_block, seeded_state
    H, Z = 128, 64
    sizes = {64: [48, 40, 33, 21, 12, 30, 5, 37], 126: [40, 26, 38, 9, 31, 17, 2, 35],
             128: [39, 3, 28, 36, 14, 40, 22, 8]}
    for N, Rs in sizes.items():
        seed = 9000 + N
        rng = np.random.default_rng(seed)
        blocks = []
        for R in Rs:
            p = min(1.0, 3.8 / max(R - 1, 1))
            up = np.triu((rng.random((R, R)) < p).astype(np.uint8), 1)
            blocks.append(up + up.T)
        graphs = [graph_from_block(b, N) for b in blocks]
        for g in graphs:
            g.num_real_nodes = torch.tensor(g.num_real_nodes, dtype=torch.long)
        model = ref.GraphVAE(1, H, Z, N)
        state0 = seeded_state(1, H, Z, N, seed)
        pos_e = sum(float(np.triu(b, 1).sum()) for b in blocks)
        pairs = sum(R * (R - 1) / 2 for R in Rs)
        rho = pos_e / (pairs + 1e-8)
        pw = float(np.clip((1 - rho) / (rho + 1e-8), 1.0, 20.0))
        state0["pos_weight_global"] = torch.tensor(pw)
        model.load_state_dict(state0)
        model.rho_global = rho
        gen = torch.Generator().manual_seed(seed)
        eps = [torch.randn(1, Z, generator=gen) for _ in graphs]
        iters = []
        orig_mpm = model._mpm

        def counting_mpm(X0, S, *a, **k):
            with IterSpy() as spy:
                X = orig_mpm(X0, S, *a, **k)
            iters.append(spy.count)
            return X

        model._mpm = counting_mpm
        res, perms = run_reference_step(ref, model, graphs, eps, train=True)
        res["loss"].backward()
        store = {"meta": np.array(f"enzymes-shaped/deg max_nodes={N}; encoder parity unpinned (PyG GCNConv restated)"),
                 "N": np.int64(N), "H": np.int64(H), "Z": np.int64(Z), "seed": np.int64(seed),
                 "rho": np.float64(rho), "pos_weight": np.float64(pw), "n_graphs": np.int64(len(graphs))}
        for i, b in enumerate(blocks):
            store[f"g{i}/block"] = b
            store[f"g{i}/eps"] = eps[i].numpy()
            store[f"g{i}/perm"] = perms[i]
            store[f"g{i}/P"] = res["P"][i]
            store[f"g{i}/X"] = res["X"][i]
            store[f"g{i}/iters"] = np.int64(iters[i])
        for k in ("rec", "kl", "loss", "max_logit", "logvar_max", "rec_i", "kl_i"):
            store[f"out/{k}"] = res[k].detach().numpy()
        for n_, p in model.named_parameters():
            store[f"grad/{n_}"] = p.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running" in k or "num_batches" in k:
                store[f"state1/{k}"] = v.numpy().copy()
        with torch.no_grad():
            res_e, _ = run_reference_step(ref, model, graphs, eps, train=False)
        for k in ("rec", "kl", "loss"):
            store[f"out_eval/{k}"] = res_e[k].numpy()
        np.savez_compressed(out / f"g11_forward_N{N}.npz", **store)
        print(f"G11 N={N}: loss={res['loss'].item():.6f} rec={res['rec'].item():.6f} kl={res['kl'].item():.6f} "
              f"iters={iters[:len(graphs)]}")


def gen_rho(ref, out: Path):
    """precompute_dataset_rho (reference model.py:99-118) on the grid graphs."""
    nxg, N = graphs_for_forward("grid")
    graphs = [nx_to_graph(g, N, "deg") for g in nxg]
    for g in graphs:
        g.to = lambda dev, _g=g: _g
    model = ref.GraphVAE(1, 8, 4, N)
    model.precompute_dataset_rho(graphs)
    np.savez(out / "g8_rho.npz", rho=np.float64(model.rho_global),
             pos_weight=model.pos_weight_global.numpy())


def gen_sampler(ref, out: Path):
    """The reference sampler (experiments/sample.py `main`) run on a seeded checkpoint: the
    graphs it writes, for two threshold pairs, with the latent vectors it drew and the
    decoder logits of the same model."""
    import contextlib
    import importlib.util
    import io
    import tempfile

    import networkx as nx
    import yaml

    spec = importlib.util.spec_from_file_location("ref_sample", REF_ROOT + "/experiments/sample.py")
    smod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(smod)
    F, H, Z, N, num, seed = 1, 16, 8, 12, 6, 3
    torch.manual_seed(7)
    model = ref.GraphVAE(F, H, Z, N)
    with torch.no_grad():                      # spread the logits so that both thresholds bite
        for name, p in model.named_parameters():
            if name.startswith("dec_"):
                p.mul_(2.5) if p.dim() == 2 else p.uniform_(-0.5, 0.5)
    store = {"dims": np.array([F, H, Z, N, num, seed], dtype=np.int64)}
    for k, v in model.state_dict().items():
        store[f"state/{k}"] = v.detach().numpy().copy()
    with tempfile.TemporaryDirectory() as td:
        run = Path(td)
        torch.save(model.state_dict(), run / "graphvae_ddp.pt")
        yaml.safe_dump({"in_dim": F, "hid_dim": H, "z_dim": Z, "max_nodes": N, "pool": "sum"},
                       open(run / "params.yaml", "w"))
        for tag, (th_e, th_n) in (("a", (0.5, 0.5)), ("b", (0.3, 0.6))):
            drawn = []
            orig = torch.randn

            def spy(*a, **k):
                t = orig(*a, **k)
                drawn.append(t.detach().clone())
                return t

            torch.randn = spy
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    smod.main(run, num, th_e, th_n, "cpu", seed)
            finally:
                torch.randn = orig
            z = [t for t in drawn if tuple(t.shape) == (num, Z)][-1]
            store[f"{tag}/th"] = np.array([th_e, th_n], dtype=np.float64)
            store[f"{tag}/z"] = z.numpy()
            n_edges = 0
            for i in range(num):
                g = nx.read_gml(run / "samples_fixed" / f"sample_{i}.gml", label="label")
                store[f"{tag}/nodes{i}"] = np.array(sorted(int(v) for v in g.nodes()), dtype=np.int64)
                e = sorted((min(int(u), int(v)), max(int(u), int(v))) for u, v in g.edges())
                store[f"{tag}/edges{i}"] = np.array(e, dtype=np.int64).reshape(-1, 2)
                n_edges += len(e)
            with torch.no_grad():
                store[f"{tag}/edge_logits"] = model.dec_edge(z).numpy()
                store[f"{tag}/node_logits"] = model.dec_node(z).numpy()
            print(f"G9 sampler {tag}: th=({th_e},{th_n}) {n_edges} edges over {num} samples")
    np.savez_compressed(out / "g9_sampler.npz", **store)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(Path(__file__).resolve().parents[1] / "tests" / "golden"))
    ap.add_argument("--only", default=None, help="regenerate one fixture family only (sampler | forward_old | forward_large | individual_rho)")
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    ref = import_reference()
    if args.only == "sampler":
        gen_sampler(ref, out)
        return
    if args.only == "forward_old":
        gen_forward_old(ref, out)
        return
    if args.only == "forward_large":
        gen_forward_large(ref, out)
        return
    if args.only == "individual_rho":
        gen_individual_rho(out)
        return
    gen_forward_test(ref, out)
    gen_reduction_orders(out)
    gen_lsap(out)
    gen_rho(ref, out)
    gen_forward(ref, out)
    gen_forward_old(ref, out)
    gen_forward_large(ref, out)
    gen_individual_rho(out)
    gen_matching(ref, out)
    gen_sampler(ref, out)


if __name__ == "__main__":
    main()
