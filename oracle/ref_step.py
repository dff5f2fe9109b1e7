"""CPU restatement of the reference GraphVAE training step, one graph at a time.

TEST INFRASTRUCTURE ONLY (checker for tests/ and smoke(), and the timed
`cpu_baseline` of bench.py); the product package never imports it.

It follows the reference's per-graph algorithm with PyTorch CPU ops and autograd:
  forward               graphvae/models/model.py:281-373
  _edge_sim_tensor      graphvae/models/model.py:171-216   (S materialised, (R,R,R,R))
  _mpm                  graphvae/models/model.py:247-276   (early stop checked per iteration)
  assignment            graphvae/models/model.py:319-323   (scipy.optimize.linear_sum_assignment)
  step loop             graphvae/train/train_ddp.py:282-292,316
GCNConv is third-party (torch_geometric 2.6.1, absent from the reference tree); it is
restated densely from its documented defaults, so encoder parity is "unpinned" by the
reference's own tests.  Pinned by tests/golden/g5_forward_*.npz (losses, per-graph
permutations, all parameter gradients, BatchNorm running statistics, three optimiser
steps), which tools/make_golden.py produced from the imported reference.

The matching block runs under the same PyTorch CPU reduction kernels as the
reference, so on an x86 host it reproduces the reference's bits; the
machine-independent bit-exact statement of those reductions is oracle/gvae_oracle.c.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import scipy.optimize
import torch
import torch.nn.functional as F

PARAM_NAMES = [
    "conv1.bias", "conv1.lin.weight", "bn1.weight", "bn1.bias",
    "conv2.bias", "conv2.lin.weight", "bn2.weight", "bn2.bias",
    "mu.weight", "mu.bias", "logvar.weight", "logvar.bias",
    "dec_edge.0.weight", "dec_edge.0.bias", "dec_edge.2.weight", "dec_edge.2.bias",
    "dec_node.0.weight", "dec_node.0.bias", "dec_node.2.weight", "dec_node.2.bias",
]
BUFFER_NAMES = ["bn1.running_mean", "bn1.running_var", "bn1.num_batches_tracked",
                "bn2.running_mean", "bn2.running_var", "bn2.num_batches_tracked",
                "pos_weight_global"]


def param_shapes(in_dim: int, hid: int, z: int, n: int) -> dict:
    u = n * (n - 1) // 2
    return {
        "conv1.bias": (hid,), "conv1.lin.weight": (hid, in_dim), "bn1.weight": (hid,), "bn1.bias": (hid,),
        "conv2.bias": (hid,), "conv2.lin.weight": (hid, hid), "bn2.weight": (hid,), "bn2.bias": (hid,),
        "mu.weight": (z, hid), "mu.bias": (z,), "logvar.weight": (z, hid), "logvar.bias": (z,),
        "dec_edge.0.weight": (hid, z), "dec_edge.0.bias": (hid,),
        "dec_edge.2.weight": (u, hid), "dec_edge.2.bias": (u,),
        "dec_node.0.weight": (hid, z), "dec_node.0.bias": (hid,),
        "dec_node.2.weight": (n, hid), "dec_node.2.bias": (n,),
    }


def init_state(in_dim: int, hid: int, z: int, n: int, seed: int = 42) -> dict:
    """Xavier-uniform (ReLU gain) weights, zero biases, BN gamma=1 beta=0
    (reference model.py:123-134)."""
    g = torch.Generator().manual_seed(seed)
    st = {}
    gain = math.sqrt(2.0)
    for name, shp in param_shapes(in_dim, hid, z, n).items():
        if name.endswith("weight") and len(shp) == 2:
            bound = gain * math.sqrt(6.0 / (shp[0] + shp[1]))
            st[name] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        elif name in ("bn1.weight", "bn2.weight"):
            st[name] = torch.ones(shp)
        else:
            st[name] = torch.zeros(shp)
    for b in ("bn1", "bn2"):
        st[f"{b}.running_mean"] = torch.zeros(hid)
        st[f"{b}.running_var"] = torch.ones(hid)
        st[f"{b}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    st["pos_weight_global"] = torch.tensor(1.0)
    return st


# ----------------------------------------------------------------------------- matching
def edge_sim_tensor(A: torch.Tensor, B: torch.Tensor, all_rows: bool = False) -> torch.Tensor:
    """reference model.py:171-216; all_rows: the older form without the cut to the real nodes
    (model_legacy.py:141-190), otherwise the same expression"""
    R = A.size(0) if all_rows else int((A.sum(1) > 0).sum())
    A = A[:R, :R].clone()
    B = B[:R, :R].clone()
    A.diagonal().fill_(1.0)
    B.diagonal().fill_(1.0)
    degA, degB = A.sum(1), B.sum(1)
    node_sim = 1.0 / ((degA.view(-1, 1) - degB.view(1, -1)).abs() + 1.0)
    dA, dB = A.diag(), B.diag()
    S = (A[:, :, None, None] * dA.view(1, R, 1, 1) * dA.view(R, 1, 1, 1)
         * B[None, None, :, :] * dB.view(1, 1, R, 1) * dB.view(1, 1, 1, R))
    eye = torch.eye(R, dtype=torch.bool)
    S[eye[:, :, None, None] ^ eye[None, None, :, :]] = 0
    idx = torch.arange(R)
    S[idx[:, None], idx[:, None], idx[None, :], idx[None, :]] = dA.view(R, 1) * dB.view(1, R) * node_sim
    return S


def mpm(X0: torch.Tensor, S: torch.Tensor, iters: int = 50, tol: float = 1e-4,
        early_stop: bool = True):
    """reference model.py:247-276; also returns the iterations executed."""
    X = X0.clone()
    N = X.size(0)
    idx = torch.arange(N)
    node_sim = S[idx[:, None], idx[:, None], idx[None, :], idx[None, :]]
    mask = (~torch.eye(N, dtype=torch.bool)).unsqueeze(-1)
    done = 0
    for _ in range(iters):
        X_prev = X
        XS_max = (S * X.unsqueeze(0).unsqueeze(2)).amax(-1)
        X = X * node_sim + (XS_max * mask).sum(dim=1)
        X = X / X.norm(p=2)
        done += 1
        if early_stop and (X - X_prev).abs().max() < tol:
            break
    return X, done


def assign_perm(X: torch.Tensor) -> torch.Tensor:
    """reference model.py:319-321"""
    row, col = scipy.optimize.linear_sum_assignment((-X).detach().numpy())
    perm = torch.empty(len(col), dtype=torch.long)
    perm[torch.as_tensor(col)] = torch.as_tensor(row)
    return perm


# ----------------------------------------------------------------------------- network
def gcn_dense(x, adj_block_real, n_real, W, b):
    """PyG GCNConv defaults on x's rows: self loop on every row, symmetric norm."""
    n = x.size(0)
    A = torch.zeros(n, n)
    r = min(n_real, n)
    A[:r, :r] = (adj_block_real[:r, :r] != 0).float()
    A.fill_diagonal_(1.0)
    dinv = A.sum(1).pow(-0.5)
    Ahat = dinv.view(-1, 1) * A * dinv.view(1, -1)
    return Ahat @ (x @ W.t()) + b


def vec_to_adj(vec, N):
    adj = vec.new_zeros(N, N)
    idx = torch.triu_indices(N, N, offset=1)
    adj[idx[0], idx[1]] = vec
    return adj + adj.T


def forward_graph(st: dict, g, eps: torch.Tensor, train: bool, bn_momentum: float = 0.1,
                  mpm_iters: int = 50, tol: float = 1e-4, early_stop: bool = True,
                  no_grad_matching: bool = False, legacy_mask: bool = False,
                  sfree_matching: bool = False, skip_matching: bool = False,
                  individual_rho: bool = False):
    """One graph through reference model.py:288-366.  `g` has x (n_x,F), adj_dense (N,N),
    num_real_nodes.  Mutates the BN running statistics in `st` when train.
    `legacy_mask` selects the older loss `forward_old` (model.py:403-408,425-427): the logits
    of padded nodes are overwritten with -10 before the node loss, so they contribute the
    constant softplus(-10) each and receive no gradient; everything else is unchanged (the
    R x R block handed to the matching is not touched by the overwrite).
    `sfree_matching`: run the MPM through the C oracle's S-free form (oracle/gvae_oracle.c,
    bit-equal to the materialised form, tests/test_oracle_golden.py) without autograd: what
    makes graphs of 64-128 nodes checkable in seconds (S would be 64 MiB - 1 GiB each).
    `skip_matching`: identity permutation (for float64 yardsticks at sizes where S does not fit:
    prediction and truth are gathered through the same permutation, model.py:325-326, so the
    loss and its gradients do not depend on it beyond rounding).
    `individual_rho`: the variant `forward_individual_rho` of model_legacy.py:257-363 on the
    current heads (its decoder `dec` is `dec_edge`; `dec_node` takes no part): matching over
    all N rows of the padded matrices, the ground truth permuted and the prediction left in
    place, positives weighted by the graph's own density among the first R permuted rows."""
    N = g.adj_dense.size(0)
    R = int(g.num_real_nodes)
    A_gt = g.adj_dense

    def bn(h, name):
        if train:
            mean = h.mean(0)
            var = h.var(0, unbiased=False)
            with torch.no_grad():
                n = h.size(0)
                st[f"{name}.running_mean"].mul_(1 - bn_momentum).add_(bn_momentum * mean)
                st[f"{name}.running_var"].mul_(1 - bn_momentum).add_(bn_momentum * var * n / (n - 1))
                st[f"{name}.num_batches_tracked"] += 1
        else:
            mean, var = st[f"{name}.running_mean"], st[f"{name}.running_var"]
        return (h - mean) / torch.sqrt(var + 1e-5) * st[f"{name}.weight"] + st[f"{name}.bias"]

    h = F.relu(bn(gcn_dense(g.x, A_gt, R, st["conv1.lin.weight"], st["conv1.bias"]), "bn1"))
    h = F.relu(bn(gcn_dense(h, A_gt, R, st["conv2.lin.weight"], st["conv2.bias"]), "bn2"))
    g_vec = h.sum(0, keepdim=True) / R
    mu = g_vec @ st["mu.weight"].t() + st["mu.bias"]
    logvar = (g_vec @ st["logvar.weight"].t() + st["logvar.bias"]).clamp(-4.0, 4.0)
    z = mu + eps * torch.exp(0.5 * logvar)
    he = F.relu(z @ st["dec_edge.0.weight"].t() + st["dec_edge.0.bias"])
    vec_logits = (he @ st["dec_edge.2.weight"].t() + st["dec_edge.2.bias"]).squeeze(0)
    hn = F.relu(z @ st["dec_node.0.weight"].t() + st["dec_node.0.bias"])
    node_logits = (hn @ st["dec_node.2.weight"].t() + st["dec_node.2.bias"]).squeeze(0)
    A_hat = vec_to_adj(vec_logits, N)
    A_hat = A_hat + torch.diag(node_logits)

    if individual_rho:
        A_hat = vec_to_adj(vec_logits, N)
        A_hat = A_hat - 10.0 * torch.eye(N)                      # model_legacy.py:291
        with torch.no_grad():                                    # :294-310 (X is used detached)
            S = edge_sim_tensor(A_gt, A_hat.sigmoid(), all_rows=True)
            X, done = mpm(torch.full((N, N), 1 / N), S, mpm_iters, tol, early_stop)
        perm = assign_perm(X)
        A_gt_p = A_gt[perm][:, perm]                             # :313-315: prediction not permuted
        tri = torch.triu_indices(R, R, offset=1)                 # pairs i < j < R, row-major (:324-331)
        truth, pred = A_gt_p[tri[0], tri[1]], A_hat[tri[0], tri[1]]
        rho = truth.sum() / (R * (R - 1) / 2 + 1e-8)             # :336-343
        pos_weight = ((1 - rho) / (rho + 1e-8)).clamp(min=1.0, max=20)
        rec = F.binary_cross_entropy_with_logits(pred, truth, pos_weight=pos_weight)
        kl = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
        return SimpleNamespace(rec=rec, kl=kl, max_logit=vec_logits.detach().abs().max(),
                               logvar_max=logvar.max().detach(), perm=perm, X=X.detach(),
                               iters_done=done, vec_logits=vec_logits.detach(),
                               node_logits=node_logits.detach(), P=A_hat.sigmoid().detach(),
                               pos_weight=pos_weight.detach())

    def matching():
        S = edge_sim_tensor(A_gt[:R, :R], A_hat.sigmoid()[:R, :R])
        X0 = torch.full((R, R), 1 / R)
        X, done = mpm(X0, S, mpm_iters, tol, early_stop)
        return X, done
    if skip_matching:
        X, done = torch.eye(R, dtype=A_hat.dtype), 0
    elif sfree_matching:
        from . import matching as ora
        Xn, done = ora.mpm_sfree(A_gt[:R, :R].numpy(), A_hat.sigmoid()[:R, :R].detach().numpy(),
                                 iters=mpm_iters, tol=tol, early_stop=early_stop)
        X = torch.from_numpy(Xn)
    elif no_grad_matching:
        with torch.no_grad():
            X, done = matching()
    else:
        X, done = matching()
    perm = assign_perm(X)
    perm_full = torch.arange(N)
    perm_full[:R] = perm
    A_gt_p = A_gt[perm_full][:, perm_full]
    A_pred_p = A_hat[perm_full][:, perm_full]
    tri = torch.triu_indices(R, R, offset=1)
    loss_edges = F.binary_cross_entropy_with_logits(A_pred_p[tri[0], tri[1]], A_gt_p[tri[0], tri[1]],
                                                    pos_weight=st["pos_weight_global"])
    node_truth = torch.zeros(N)
    node_truth[:R] = 1.0
    pos_w_node = torch.tensor((N - R) / (R + 1e-8)).clamp(1.0, 5.0)
    node_in = torch.cat([node_logits[:R], torch.full((N - R,), -10.0)]) if legacy_mask else node_logits
    loss_nodes = F.binary_cross_entropy_with_logits(node_in, node_truth, pos_weight=pos_w_node)
    node_masked = node_logits.detach().clone()
    node_masked[R:] = -10.0
    max_logit = torch.cat([vec_logits.detach(), node_masked]).abs().max()
    kl = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    return SimpleNamespace(rec=loss_edges + loss_nodes, kl=kl, max_logit=max_logit,
                           logvar_max=logvar.max().detach(), perm=perm_full, X=X.detach(),
                           iters_done=done, vec_logits=vec_logits.detach(),
                           node_logits=node_logits.detach(), P=A_hat.sigmoid().detach())


def forward_batch(st: dict, graphs, eps_list, train: bool = True, **kw):
    outs = [forward_graph(st, g, e, train, **kw) for g, e in zip(graphs, eps_list)]
    rec = torch.stack([o.rec for o in outs]).mean()
    kl = torch.stack([o.kl for o in outs]).mean()
    return SimpleNamespace(rec=rec, kl=kl, loss=rec + kl,
                           max_logit=torch.stack([o.max_logit for o in outs]).max(),
                           logvar_max=outs[-1].logvar_max, per_graph=outs)


def make_trainable(st: dict) -> dict:
    out = {}
    for k, v in st.items():
        if k in PARAM_NAMES:
            out[k] = v.clone().requires_grad_(True)
        else:
            out[k] = v.clone()
    return out


def train_step(st: dict, opt, graphs, eps_list, global_step: int, anneal_steps: int, lr: float,
               beta0: float, beta_final: float, clip_max: float, **kw):
    """One iteration of reference train_ddp.py:279-319 (single process)."""
    res = forward_batch(st, graphs, eps_list, train=True, **kw)
    frac = min(1.0, global_step / anneal_steps)
    beta = beta0 + (beta_final - beta0) * frac
    loss = res.rec + beta * res.kl
    for pg in opt.param_groups:
        pg["lr"] = lr * frac
    opt.zero_grad(set_to_none=True)
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_([st[n] for n in PARAM_NAMES], clip_max)
    opt.step()
    return loss.item(), res, float(gn)


def graphs_from_arrays(x, adj, n_x, R):
    """Per-graph inputs from dense batch arrays (numpy or torch, CPU)."""
    out = []
    for g in range(len(R)):
        xg = torch.as_tensor(np.asarray(x[g]), dtype=torch.float32)[: int(n_x[g])]
        out.append(SimpleNamespace(x=xg, adj_dense=torch.as_tensor(np.asarray(adj[g]), dtype=torch.float32),
                                   num_real_nodes=int(R[g])))
    return out
