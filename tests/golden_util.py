"""Helpers to load the g5_forward_* golden fixtures."""
from types import SimpleNamespace

import numpy as np
import torch


def load_g5(path):
    g = np.load(path, allow_pickle=False)
    n = int(g["n_graphs"])
    graphs, eps, perms, perms_eval, Ps, Xs = [], [], [], [], [], []
    for i in range(n):
        graphs.append(SimpleNamespace(
            x=torch.from_numpy(g[f"g{i}/x"]), edge_index=torch.from_numpy(g[f"g{i}/edge_index"]),
            adj_dense=torch.from_numpy(g[f"g{i}/adj_dense"]), num_real_nodes=int(g[f"g{i}/R"])))
        eps.append(torch.from_numpy(g[f"g{i}/eps"]))
        perms.append(g[f"g{i}/perm"])
        perms_eval.append(g[f"g{i}/perm_eval"])
        Ps.append(g[f"g{i}/P"])
        Xs.append(g[f"g{i}/X"])
    state0 = {k[len("state0/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state0/")}
    grads = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    state1 = {k[len("state1/"):]: g[k] for k in g.files if k.startswith("state1/")}
    out = {k[len("out/"):]: g[k] for k in g.files if k.startswith("out/")}
    out_eval = {k[len("out_eval/"):]: g[k] for k in g.files if k.startswith("out_eval/")}
    opt = {"hparams": g["opt/hparams"], "losses": g["opt/losses"],
           "state3": {k[len("opt/state3/"):]: g[k] for k in g.files if k.startswith("opt/state3/")},
           "eps": [[torch.from_numpy(g[f"opt/eps{s}_{i}"]) for i in range(n)] for s in range(3)]}
    dims = dict(N=int(g["N"]), H=int(g["H"]), Z=int(g["Z"]), in_dim=int(g["in_dim"]))
    return SimpleNamespace(graphs=graphs, eps=eps, perms=perms, perms_eval=perms_eval, P=Ps, X=Xs,
                           state0=state0,
                           grads=grads, state1=state1, out=out, out_eval=out_eval, opt=opt, dims=dims,
                           rho=float(g["rho"]))


def load_g10(golden_dir):
    """The `forward_old` fixture (outputs only) with the g5 rand/deg case it was generated on."""
    G = load_g5(golden_dir / "g5_forward_rand_deg.npz")
    g = np.load(golden_dir / "g10_forward_old_rand_deg.npz", allow_pickle=False)
    out = {k[len("out/"):]: g[k] for k in g.files if k.startswith("out/")}
    grads = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    perms = [g[f"g{i}/perm"] for i in range(len(G.graphs))]
    return G, SimpleNamespace(out=out, grads=grads, perms=perms)


def load_g12(golden_dir):
    """The `forward_individual_rho` fixture (model_legacy.py:257-363) with the g5 rand/id case it
    was generated on."""
    G = load_g5(golden_dir / "g5_forward_rand_id.npz")
    g = np.load(golden_dir / "g12_individual_rho_rand_id.npz", allow_pickle=False)
    out = {k[len("out/"):]: g[k] for k in g.files if k.startswith("out/")}
    grads = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    n = len(G.graphs)
    return G, SimpleNamespace(out=out, grads=grads, perms=[g[f"g{i}/perm"] for i in range(n)],
                              P=[g[f"g{i}/P"] for i in range(n)], X=[g[f"g{i}/X"] for i in range(n)])


PARAM_SHAPES = lambda F, H, Z, N: {  # noqa: E731  (reference model.py:71-89, named_parameters order)
    "conv1.bias": (H,), "conv1.lin.weight": (H, F), "bn1.weight": (H,), "bn1.bias": (H,),
    "conv2.bias": (H,), "conv2.lin.weight": (H, H), "bn2.weight": (H,), "bn2.bias": (H,),
    "mu.weight": (Z, H), "mu.bias": (Z,), "logvar.weight": (Z, H), "logvar.bias": (Z,),
    "dec_edge.0.weight": (H, Z), "dec_edge.0.bias": (H,),
    "dec_edge.2.weight": (N * (N - 1) // 2, H), "dec_edge.2.bias": (N * (N - 1) // 2,),
    "dec_node.0.weight": (H, Z), "dec_node.0.bias": (H,),
    "dec_node.2.weight": (N, H), "dec_node.2.bias": (N,)}


def seeded_state(F: int, H: int, Z: int, N: int, seed: int) -> dict:
    """A full state_dict as a pure function of (dims, seed): Xavier-uniform (gain sqrt 2) weight
    matrices, biases and BatchNorm affine terms perturbed by 0.1 * normal so that every gradient
    path is exercised, fresh running statistics.  Fixtures generated from large models store
    the seed instead of megabytes of weights; generator and tests both call this."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shp in PARAM_SHAPES(F, H, Z, N).items():
        if len(shp) == 2:
            b = np.sqrt(2.0) * np.sqrt(6.0 / (shp[0] + shp[1]))
            a = rng.uniform(-b, b, shp)
        elif name in ("bn1.weight", "bn2.weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        else:
            a = 0.1 * rng.standard_normal(shp)
        st[name] = torch.from_numpy(a.astype(np.float32))
    for b in ("bn1", "bn2"):
        st[f"{b}.running_mean"] = torch.zeros(H)
        st[f"{b}.running_var"] = torch.ones(H)
        st[f"{b}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    st["pos_weight_global"] = torch.tensor(1.0)
    return st


def graph_from_block(block: np.ndarray, N: int):
    """Graph record of the reference's data contract (train_ddp.py:67-112, `deg` features) from
    the R x R 0/1 adjacency block without self loops."""
    R = block.shape[0]
    adj = torch.zeros(N, N)
    adj[:R, :R] = torch.from_numpy(block.astype(np.float32))
    adj[torch.arange(R), torch.arange(R)] = 1.0
    x = torch.zeros(N, 1)
    x[:R, 0] = torch.from_numpy(block.sum(1).astype(np.float32))
    src, dst = np.nonzero(block)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    return SimpleNamespace(x=x, edge_index=ei, adj_dense=adj, num_real_nodes=R)


def load_g11(path):
    """g11_forward_N*.npz: the reference's forward/backward at max_nodes 64 / 126 / 128 with the
    trainer's default widths, on ENZYMES-shaped graphs small enough for the reference's CPU path."""
    g = np.load(path, allow_pickle=False)
    n, N, H, Z, seed = (int(g[k]) for k in ("n_graphs", "N", "H", "Z", "seed"))
    graphs = [graph_from_block(g[f"g{i}/block"], N) for i in range(n)]
    state0 = seeded_state(1, H, Z, N, seed)
    state0["pos_weight_global"] = torch.tensor(float(g["pos_weight"]))
    grads = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    return SimpleNamespace(
        graphs=graphs, eps=[torch.from_numpy(g[f"g{i}/eps"]) for i in range(n)],
        perms=[g[f"g{i}/perm"] for i in range(n)], P=[g[f"g{i}/P"] for i in range(n)],
        X=[g[f"g{i}/X"] for i in range(n)], iters=[int(g[f"g{i}/iters"]) for i in range(n)],
        state0=state0, grads=grads,
        state1={k[len("state1/"):]: g[k] for k in g.files if k.startswith("state1/")},
        out={k[len("out/"):]: g[k] for k in g.files if k.startswith("out/")},
        out_eval={k[len("out_eval/"):]: g[k] for k in g.files if k.startswith("out_eval/")},
        dims=dict(N=N, H=H, Z=Z, in_dim=1), rho=float(g["rho"]))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


ZERO_GRAD_PARAMS = ("conv1.bias", "conv2.bias")


def truth64(G, want_res: bool = False, **kw):
    """Loss and gradients of the golden case recomputed in float64 by the CPU
    restatement: the yardstick for how far the fp32 reference itself is from the exact
    answer (BatchNorm over a handful of near-identical rows is ill-conditioned)."""
    from oracle import ref_step as R
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        st = R.make_trainable({k: (v.double() if v.is_floating_point() else v.clone())
                               for k, v in G.state0.items()})
        graphs = [SimpleNamespace(x=g.x.double(), adj_dense=g.adj_dense.double(),
                                  num_real_nodes=g.num_real_nodes) for g in G.graphs]
        res = R.forward_batch(st, graphs, [e.double() for e in G.eps], train=True, **kw)
        res.loss.backward()
        grads = {n: (st[n].grad.numpy().copy() if st[n].grad is not None
                     else np.zeros(tuple(st[n].shape))) for n in R.PARAM_NAMES}
        return (grads, res) if want_res else grads
    finally:
        torch.set_default_dtype(old)


# conv1.lin.weight is an (H, F) sum over every node row of the batch with heavy cancellation (for
# `deg` features the terms are degree-weighted): two fp32 evaluations of it differ by several
# times the distance of either from the float64 value (measured: two CPU evaluations, or CPU and
# GPU, land on opposite sides of it).
CANCELLATION_PARAMS = ("conv1.lin.weight",)


def check_grads(got: dict, ref: dict, truth: dict | None = None, tol: float = 1e-5,
                slack_factor: float = 2.0):
    """Each gradient tensor within `tol` relative (to its largest entry) of the
    reference's, plus, when `truth` (float64 recomputation) is given, `slack_factor` times the
    reference's own distance from it: two fp32 evaluations that are equally accurate sit up to
    twice that distance apart (opposite sides of the exact value), so 2 is the smallest factor
    that does not ask the candidate to be MORE accurate than the reference (4 for the
    cancellation-dominated tensor above).  Prints, per tensor, the distances of both from the
    float64 value (pytest -s shows them).  The two GCN biases feed
    straight into a training-mode BatchNorm, which cancels any per-column constant: their
    true gradient is zero and the reference holds only rounding noise there, so those are
    only required to be noise (<= 20*tol of the model-wide gradient scale; the reference's
    own noise reaches 5*tol of it on the grid graphs)."""
    gmax = max(float(np.abs(v).max()) for v in ref.values())
    for name, r in ref.items():
        g = np.asarray(got[name], np.float64)
        r = np.asarray(r, np.float64)
        if name in ZERO_GRAD_PARAMS:
            assert np.abs(g).max() <= 20 * tol * gmax and np.abs(r).max() <= 20 * tol * gmax, name
            continue
        k = 2.0 * slack_factor if name in CANCELLATION_PARAMS else slack_factor
        slack = k * float(np.abs(r - truth[name]).max()) if truth is not None else 0.0
        err = float(np.abs(g - r).max())
        if truth is not None:
            m = max(float(np.abs(truth[name]).max()), 1e-30)
            print(f"  grad {name:20s} |got-ref| {err / m:.2e}  |got-f64| {np.abs(g - truth[name]).max() / m:.2e}  "
                  f"|ref-f64| {np.abs(r - truth[name]).max() / m:.2e}")
        assert err <= tol * max(np.abs(r).max(), 1e-30) + slack, (name, err, float(np.abs(r).max()), slack)


def check_grads_accurate(got: dict, ref: dict, truth: dict, tol: float = 1e-5):
    """The bar of the GPU step: each gradient tensor is within `tol` (relative to the tensor's
    largest entry) of the float64 value, or at least as close to it as the reference's own fp32
    evaluation is (`tol` + the reference's distance).  The reference's fp32 gradients sit up to
    3e-4 from the float64 value on ill-conditioned BatchNorm columns; the GPU computes the encoder
    forward in fp64 and sits 1e-7 .. 1e-6 from it there.  Also required: within `tol` + twice the
    reference's distance of the reference itself (implied by the first, kept as a cross-check).
    The two GCN biases have a true gradient of zero (a training-mode BatchNorm cancels any
    per-column constant) and are only required to be noise, as in check_grads."""
    gmax = max(float(np.abs(v).max()) for v in ref.values())
    for name, r in ref.items():
        g, r, t = (np.asarray(a, np.float64) for a in (got[name], r, truth[name]))
        if name in ZERO_GRAD_PARAMS:
            assert np.abs(g).max() <= 20 * tol * gmax, name
            continue
        scale = max(float(np.abs(t).max()), 1e-30)
        d_got, d_ref = float(np.abs(g - t).max()), float(np.abs(r - t).max())
        print(f"  grad {name:20s} |got-f64| {d_got / scale:.2e}  |ref-f64| {d_ref / scale:.2e}  "
              f"|got-ref| {np.abs(g - r).max() / scale:.2e}")
        assert d_got <= tol * scale + d_ref, (name, d_got / scale, d_ref / scale)
        assert float(np.abs(g - r).max()) <= tol * scale + 2.0 * d_ref, name


def truth64_steps(G, n_steps: int = 3):
    """The golden optimiser steps (reference train_ddp.py:282-292,316) recomputed in float64 by
    the CPU restatement: per step (loss, gradient norm before clipping), and the parameters
    after the last step."""
    from oracle import ref_step as R
    lr, wd, beta0, beta_final, clip_max, anneal = G.opt["hparams"].tolist()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        st = R.make_trainable({k: (v.double() if v.is_floating_point() else v.clone())
                               for k, v in G.state0.items()})
        graphs = [SimpleNamespace(x=g.x.double(), adj_dense=g.adj_dense.double(),
                                  num_real_nodes=g.num_real_nodes) for g in G.graphs]
        opt = torch.optim.Adam([st[n] for n in R.PARAM_NAMES], lr=lr, weight_decay=wd)
        steps = []
        for gs in range(n_steps):
            loss, _res, gn = R.train_step(st, opt, graphs, [e.double() for e in G.opt["eps"][gs]], gs,
                                          int(anneal), lr, beta0, beta_final, clip_max)
            steps.append((loss, gn))
        return steps, {k: st[k].detach().numpy().copy() for k in R.PARAM_NAMES}
    finally:
        torch.set_default_dtype(old)


def perm_equivalent(perm, perm_ref, X, tol=1e-5):
    """True if `perm` equals the reference permutation, or both are optimal for X up to
    rounding (ties between assignments of near-identical value: an end-to-end run reaches
    the matching with probabilities that differ from the reference's in the last bits, and
    on graphs with symmetries that can flip the choice between equally good assignments).
    perm[a] = row assigned to column a; objective = sum_a X[perm[a], a]."""
    perm, perm_ref = np.asarray(perm), np.asarray(perm_ref)
    if np.array_equal(perm, perm_ref):
        return True
    r = len(perm_ref)
    X = np.asarray(X, np.float64)[:r, :r]
    cols = np.arange(r)
    v, vr = X[perm[:r], cols].sum(), X[perm_ref, cols].sum()
    return sorted(perm[:r].tolist()) == list(range(r)) and abs(v - vr) <= tol * abs(vr)


def check_params_after_adam(got: dict, ref: dict, lr: float, tol: float = 1e-4):
    """Parameters after the three golden optimiser steps.  Adam divides each gradient by
    its own running magnitude, so an entry whose gradient is pure rounding noise (the GCN
    biases, see check_grads) moves by +-lr per step in a direction set by that noise: those
    two tensors are only required to stay within 3*lr of the reference, and so are the
    BatchNorm running means, which contain those biases (the mean of conv output + bias; the
    running statistics after ONE step are held to 1e-5 by the forward tests).  Everything
    else must agree to `tol` of the tensor's scale."""
    for name, r in ref.items():
        if name not in got:
            continue
        g, r = np.asarray(got[name], np.float64), np.asarray(r, np.float64)
        if name in ZERO_GRAD_PARAMS or name.endswith("running_mean"):
            assert np.abs(g - r).max() <= 3.0 * lr * 1.01, name
        else:
            assert np.abs(g - r).max() <= tol * max(np.abs(r).max(), 1e-3), (name, float(np.abs(g - r).max()))
