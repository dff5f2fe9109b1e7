"""oracle/ref_step.py (CPU restatement of the whole step) against the golden
vectors of the imported reference: losses, per-graph permutations, every parameter
gradient, BatchNorm running statistics, and three optimiser steps.
Tolerance: 1e-5 relative (north-star bar for loss and gradients); permutations exact."""
import copy

import numpy as np
import pytest
import torch

from oracle import ref_step as R
from oracle import matching as ora
from golden_util import (check_grads, check_params_after_adam, load_g5, load_g10, load_g12,
                         perm_equivalent, rel_err, truth64)

CASES = ["grid_deg", "grid_id", "rand_deg", "rand_id"]
TOL = 1e-5
# conv1.lin.weight of the legacy-loss case is a (H, 1) tensor whose entries are sums with heavy
# cancellation once the padded node logits stop contributing: fp32 evaluations of the same
# formula differ by ~4e-5 of its scale (the reference sits 6e-6 from the float64 value).
LEGACY_GRAD_TOL = 5e-5


@pytest.mark.parametrize("case", CASES)
def test_forward_backward_matches_reference(golden_dir, case):
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    st = R.make_trainable(G.state0)
    res = R.forward_batch(st, G.graphs, G.eps, train=True)
    seen = []
    for k in ("rec", "kl", "loss", "max_logit", "logvar_max"):
        assert rel_err(getattr(res, k).detach().numpy(), G.out[k]) < TOL, k
    for o, p, Pref, Xref in zip(res.per_graph, G.perms, G.P, G.X):
        r = len(p)
        assert rel_err(o.P[:r, :r].numpy(), Pref) < TOL
        assert perm_equivalent(o.perm[:r].numpy(), p, o.X.numpy())
        # matching pinned exactly on the reference's own (A, P): C oracle == reference bits
        A = G.graphs[len(seen)].adj_dense.numpy()[:r, :r]
        Xo, _ = ora.mpm_sfree(A, Pref)
        assert np.array_equal(Xo, Xref)
        assert np.array_equal(ora.assign_perm(Xo), p)
        seen.append(r)
    assert sum(np.array_equal(o.perm[:len(p)].numpy(), p) for o, p in zip(res.per_graph, G.perms)) \
        >= len(G.perms) - 1
    res.loss.backward()
    check_grads({n: st[n].grad.numpy() for n in R.PARAM_NAMES}, G.grads, truth64(G), TOL)
    for k, v in G.state1.items():
        assert rel_err(st[k].detach().numpy(), v) < TOL, k


@pytest.mark.parametrize("case", CASES)
def test_eval_mode_matches_reference(golden_dir, case):
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    st = R.make_trainable(G.state0)
    with torch.no_grad():
        R.forward_batch(st, G.graphs, G.eps, train=True)      # moves the running stats
        res = R.forward_batch(st, G.graphs, G.eps, train=False)
    for k in ("rec", "kl", "loss"):
        assert rel_err(res.__dict__[k].numpy(), G.out_eval[k]) < TOL, k
    for o, p in zip(res.per_graph, G.perms_eval):
        assert perm_equivalent(o.perm[:len(p)].numpy(), p, o.X.numpy())


@pytest.mark.parametrize("case", CASES)
def test_three_optimizer_steps(golden_dir, case):
    """Schedule of reference train_ddp.py:282-292,316."""
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    lr, wd, beta0, beta_final, clip_max, anneal = G.opt["hparams"].tolist()
    st = R.make_trainable(G.state0)
    opt = torch.optim.Adam([st[n] for n in R.PARAM_NAMES], lr=lr, weight_decay=wd)
    for gs in range(3):
        loss, res, gn = R.train_step(st, opt, G.graphs, G.opt["eps"][gs], gs, int(anneal), lr, beta0,
                                     beta_final, clip_max)
        ref = G.opt["losses"][gs]
        assert abs(loss - ref[0]) <= 2e-5 * abs(ref[0]), (gs, loss, ref[0])
        assert abs(gn - ref[3]) <= 1e-4 * abs(ref[3]), (gs, gn, ref[3])
    check_params_after_adam({k: st[k].detach().numpy() for k in R.PARAM_NAMES}, G.opt["state3"], lr)


def test_legacy_loss_matches_reference_forward_old(golden_dir):
    """`forward_old` (reference model.py:378-440): padded node logits read as -10."""
    G, L = load_g10(golden_dir)
    st = R.make_trainable(G.state0)
    res = R.forward_batch(st, G.graphs, G.eps, train=True, legacy_mask=True)
    for k in ("rec", "kl", "max_logit", "logvar_max"):
        assert rel_err(getattr(res, k).detach().numpy(), L.out[k]) < TOL, k
    assert rel_err(torch.stack([o.rec for o in res.per_graph]).detach().numpy(), L.out["rec_i"]) < TOL
    assert abs(float(res.rec) - float(G.out["rec"])) > 0.1      # the variant is not a no-op here
    for o, p in zip(res.per_graph, L.perms):
        assert perm_equivalent(o.perm[:len(p)].numpy(), p, o.X.numpy())
    (res.rec + res.kl).backward()
    check_grads({n: st[n].grad.numpy() for n in R.PARAM_NAMES}, L.grads,
                truth64(G, legacy_mask=True), LEGACY_GRAD_TOL)


def test_individual_rho_matches_reference(golden_dir):
    """`forward_individual_rho` (reference model_legacy.py:257-363): the one variant in which the
    permutation moves the loss (only the ground truth is permuted)."""
    G, L = load_g12(golden_dir)
    st = R.make_trainable(G.state0)
    res = R.forward_batch(st, G.graphs, G.eps, train=True, individual_rho=True)
    for k in ("rec", "kl", "max_logit", "logvar_max"):
        assert rel_err(getattr(res, k).detach().numpy(), L.out[k]) < TOL, k
    assert rel_err(torch.stack([o.rec for o in res.per_graph]).detach().numpy(), L.out["rec_i"]) < TOL
    N = G.graphs[0].adj_dense.shape[0]
    for g, (o, p, Pref, Xref) in enumerate(zip(res.per_graph, L.perms, L.P, L.X)):
        assert len(p) == N and rel_err(o.P.numpy(), Pref) < TOL
        assert perm_equivalent(o.perm.numpy(), p, o.X.numpy())
        # the matching over all N rows pinned on the reference's own P: C oracle == reference bits
        A = G.graphs[g].adj_dense.numpy().copy()
        np.fill_diagonal(A, 1.0)
        Xo, _ = ora.mpm_sfree(A, Pref)
        assert np.array_equal(Xo, Xref)
        assert np.array_equal(ora.assign_perm(Xo), p)
    # a wrong permutation would show: the identity gives another loss
    assert abs(float(res.rec) - float(G.out["rec"])) > 0.1
    (res.rec + res.kl).backward()
    grads = {n: (st[n].grad.numpy() if st[n].grad is not None else np.zeros(tuple(st[n].shape), np.float32))
             for n in R.PARAM_NAMES}
    want = dict(L.grads)
    for n in R.PARAM_NAMES:                       # the node head takes no part in this loss
        if n.startswith("dec_node"):
            assert not grads[n].any()
            want[n] = grads[n]
    check_grads(grads, want, truth64(G, individual_rho=True), TOL)


@pytest.mark.parametrize("N", [64, 126, 128])
def test_large_max_nodes_matches_reference(golden_dir, N):
    """g11: the imported reference at max_nodes 64 / 126 / 128, hid 128, z 64.  The restatement
    with the materialised S and with the C oracle's S-free MPM must both reproduce it."""
    from golden_util import load_g11
    G = load_g11(golden_dir / f"g11_forward_N{N}.npz")
    for sfree in (False, True):
        st = R.make_trainable(G.state0)
        res = R.forward_batch(st, G.graphs, G.eps, train=True, no_grad_matching=True, sfree_matching=sfree)
        for k in ("rec", "kl", "loss", "max_logit", "logvar_max"):
            assert rel_err(getattr(res, k).detach().numpy(), G.out[k]) < TOL, k
        assert rel_err(torch.stack([o.rec for o in res.per_graph]).detach().numpy(), G.out["rec_i"]) < TOL
        for g, (o, p, Pref, Xref) in enumerate(zip(res.per_graph, G.perms, G.P, G.X)):
            r = len(p)
            assert rel_err(o.P[:r, :r].numpy(), Pref) < TOL
            # end to end, P differs from the reference's in the last bits (dense aggregation here,
            # scatter-add in the fixture's GCNConv), and these sparse graphs have interchangeable
            # nodes (isolated nodes, twin leaves): equal-valued assignments may swap
            assert perm_equivalent(o.perm[:r].numpy(), p, o.X.numpy())
            # the matching pinned exactly on the reference's own (A, P), with its iteration count
            Xo, done = ora.mpm_sfree(G.graphs[g].adj_dense.numpy()[:r, :r], Pref)
            assert np.array_equal(Xo, Xref) and done == G.iters[g]
            assert np.array_equal(ora.assign_perm(Xo), p)
        res.loss.backward()
        check_grads({n: st[n].grad.numpy() for n in R.PARAM_NAMES}, G.grads, truth64(G), TOL)
        for k, v in G.state1.items():
            assert rel_err(st[k].detach().numpy(), v) < TOL, k
    with torch.no_grad():
        res = R.forward_batch(st, G.graphs, G.eps, train=False, sfree_matching=True)
    for k in ("rec", "kl", "loss"):
        assert rel_err(res.__dict__[k].numpy(), G.out_eval[k]) < TOL, k
