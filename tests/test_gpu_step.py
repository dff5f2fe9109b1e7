"""GPU parity of the whole GraphVAE step (encoder, heads, decoder, matching, assignment,
losses, backward, Adam) against the golden vectors of the imported reference and against
the CPU restatement (oracle/ref_step.py).

Bars: losses / probabilities / gradients within 1e-5 relative (gradients additionally get the
reference's own distance from the float64 value, see golden_util.check_grads); the matching
run on the reference's own (A, P) must give the reference's permutation exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import (check_grads_accurate, check_params_after_adam, load_g5, load_g10,  # noqa: E402
                         load_g12, perm_equivalent, rel_err, truth64, truth64_steps)
from oracle import ref_step as R  # noqa: E402  (checker only)

CASES = ["grid_deg", "grid_id", "rand_deg", "rand_id"]
TOL = 1e-5
DEV = "cuda:0"

def _model_and_batch(G, matching="auto"):
    from graphvae.data import collate
    from graphvae.models import GraphVAE
    d = G.dims
    model = GraphVAE(d["in_dim"], d["H"], d["Z"], d["N"], matching=matching)
    missing = model.load_state_dict(G.state0, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.rho_global = G.rho
    model = model.to(DEV)
    batch = collate(G.graphs, d["N"]).to(DEV)
    eps = torch.cat(G.eps, 0).to(DEV)
    return model, batch, eps


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("matching", ["fused", "stream"])
def test_forward_backward_vs_reference_golden(golden_dir, case, matching):
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    model, batch, eps = _model_and_batch(G, matching)
    model.train()
    out = model.forward_with_eps(batch, eps)
    for k in ("rec", "kl", "loss", "max_logit", "logvar_max"):
        assert rel_err(out[k].item(), G.out[k]) < TOL, (k, out[k].item(), G.out[k])
    B = batch.num_graphs
    eng = model._engine
    assert int(eng.debug_tensor("status", B).sum()) == 0
    P = eng.debug_tensor("P", B).cpu().numpy()
    X = eng.debug_tensor("X", B).cpu().numpy()
    perm = model.last_permutations().cpu().numpy()
    n_equal = 0
    for g, (p, Pref) in enumerate(zip(G.perms, G.P)):
        r = len(p)
        off = ~np.eye(r, dtype=bool)
        assert rel_err(P[g, :r, :r][off], Pref[off]) < TOL
        assert perm_equivalent(perm[g, :r], p, X[g])
        assert np.array_equal(perm[g, r:], np.arange(r, perm.shape[1]))
        n_equal += np.array_equal(perm[g, :r], p)
    assert n_equal >= len(G.perms) - 1
    assert rel_err(eng.debug_tensor("rec_i", B).cpu().numpy(), G.out["rec_i"]) < TOL
    assert rel_err(eng.debug_tensor("kl_i", B).cpu().numpy(), G.out["kl_i"]) < TOL

    out["loss"].backward()
    got = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    check_grads_accurate(got, G.grads, truth64(G), TOL)
    sd = model.state_dict()
    for k, v in G.state1.items():
        assert rel_err(sd[k].cpu().numpy(), v) < TOL, k


@pytest.mark.parametrize("case", CASES)
def test_matching_on_reference_probabilities_is_bit_exact(golden_dir, case):
    """The matching block fed with the reference's own (A, P) of the forward pass."""
    from graphvae import _hip
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    N = G.dims["N"]
    Rh = [len(p) for p in G.perms]
    A = torch.stack([g.adj_dense for g in G.graphs]).to(DEV)
    P = torch.zeros(len(Rh), N, N)
    for g, Pref in enumerate(G.P):
        P[g, :Rh[g], :Rh[g]] = torch.from_numpy(Pref)
    P = P.to(DEV)
    Rt = torch.tensor(Rh, dtype=torch.int32, device=DEV)
    NS = _hip.node_sim(A, P, Rt)
    X = _hip.x0_uniform(Rh, N, DEV)
    _hip.mpm_fused(A, P, NS, Rt, max(Rh), X)
    perm, status = _hip.lsap(X, Rt, max(Rh))
    assert int(status.sum()) == 0
    for g, r in enumerate(Rh):
        assert np.array_equal(X[g, :r, :r].cpu().numpy(), G.X[g])
        assert np.array_equal(perm[g, :r].cpu().numpy(), G.perms[g])


@pytest.mark.parametrize("case", CASES)
def test_eval_mode_vs_reference_golden(golden_dir, case):
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    model, batch, eps = _model_and_batch(G)
    with torch.no_grad():
        model.train()
        model.forward_with_eps(batch, eps)        # moves the running statistics, as the golden run did
        model.eval()
        out = model.forward_with_eps(batch, eps)
    for k in ("rec", "kl", "loss"):
        assert rel_err(out[k].item(), G.out_eval[k]) < TOL, k
    assert int(model.bn1.num_batches_tracked) == batch.num_graphs


@pytest.mark.parametrize("case", CASES)
def test_three_optimizer_steps_vs_reference_golden(golden_dir, case):
    """Warm-up schedule, clipping and Adam of reference train_ddp.py:282-292,316."""
    from graphvae.train.train_ddp import TrainStep
    G = load_g5(golden_dir / f"g5_forward_{case}.npz")
    lr, wd, beta0, beta_final, clip_max, anneal = G.opt["hparams"].tolist()
    model, batch, _ = _model_and_batch(G)
    model.train()
    step = TrainStep(model, lr, wd, clip_max)
    t_steps, t_state = truth64_steps(G)
    for gs in range(3):
        frac = min(1.0, gs / anneal)
        beta = beta0 + (beta_final - beta0) * frac
        eps = torch.cat(G.opt["eps"][gs], 0).to(DEV)
        out4 = step(batch, beta, frac, eps=eps).tolist()
        ref = G.opt["losses"][gs]
        assert abs(out4[0] + beta * out4[1] - ref[0]) <= 2e-5 * abs(ref[0]), (gs, out4, ref)
        # Gradient norm.  Steps 0 and 1 start from the initial parameters (the warm-up makes the
        # learning rate of step 0 zero): within 1e-5 of the float64 value, or as close to it as
        # the reference is.  From step 2 on the parameters have taken an Adam update, which
        # divides each gradient entry by its own magnitude: an entry whose sign is rounding noise
        # moves by +-lr in whichever direction the noise points, in fp32 and in float64 alike, so
        # the three trajectories are 1e-5 .. 1e-4 apart whatever the accuracy of a single step.
        gn, gn_t = step.grad_norm.item(), t_steps[gs][1]
        print(f"  step {gs}: |gn-f64| {abs(gn - gn_t) / gn_t:.2e}  |ref-f64| {abs(ref[3] - gn_t) / gn_t:.2e}")
        if gs < 2:
            assert abs(gn - gn_t) <= 1e-5 * gn_t + abs(ref[3] - gn_t), (gs, gn, ref[3], gn_t)
        else:
            assert abs(gn - gn_t) <= 1e-4 * gn_t and abs(gn - ref[3]) <= 1e-4 * gn_t, (gs, gn, ref[3], gn_t)
    got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    check_params_after_adam(got, G.opt["state3"], lr, tol=1e-4)


@pytest.mark.parametrize("matching", ["fused", "stream"])
def test_overlapped_matching_equals_sequential_step(matching):
    """TrainStep with the matching half on a second stream (concurrent with backward and Adam)
    against the same steps in stream order: losses, permutations, iteration counts and the
    parameters after several steps must be bit-identical."""
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    from graphvae.train.train_ddp import TrainStep
    N, B = 32, 48
    host = [synthetic_batch("enzymes", B, N, "deg", seed=s) for s in (3, 4)]
    runs = []
    # the last schedule gets no probe at all (tries = 0): the side streams are then whatever the
    # pool hands out, possibly sharing a hardware queue with the current stream
    for overlap, pipeline, tries in ((False, False, 4), (True, False, 4), (True, True, 4), (True, True, 0)):
        torch.manual_seed(42)
        model = GraphVAE(1, 64, 32, N, matching=matching)
        model.precompute_dataset_rho(host)
        model = model.to(DEV).train()
        step = TrainStep(model, 1e-2, 1e-4, 5.0, overlap_matching=overlap, pipeline=pipeline,
                         probe_tries=tries)
        assert step.overlap_matching == overlap and step.pipeline == pipeline
        gen = torch.Generator(device="cpu").manual_seed(9)
        outs, perms, iters = [], [], []
        for s in range(6):
            batch = host[s % 2].to(DEV)
            eps = torch.randn(B, 32, generator=gen).to(DEV)
            out4 = step(batch, 0.5, 1.0, eps=eps)
            if s % 3 != 1:          # pipelined: some steps are left in flight behind the next one
                step.wait()
                outs.append(out4.cpu())
                perms.append(model.last_permutations().cpu())
                iters.append(model.last_mpm_iterations().cpu())
        loss_sum = step.pop_loss_sum().cpu()
        assert float(step.loss_sum) == 0.0
        if overlap:
            from graphvae.train.train_ddp import concurrent_stream
            assert step._side.priority == 0 and (step._tail is None or step._tail.priority == 0)
            if tries == 0:
                assert concurrent_stream.last_probe == (0, False)
        model._engine.check_matching()
        runs.append((torch.stack(outs), torch.stack(perms), torch.stack(iters), loss_sum,
                     model.flat_parameters().detach().cpu().clone(), model.bn_state().cpu().clone()))
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    assert int(runs[0][2].min()) >= 1 and float(runs[0][3]) > 0.0


def test_split_forward_phase_order_is_enforced(golden_dir):
    """The match/loss half in separate calls: the assignment cannot be requested before the
    matching, no phase twice, nothing without a model half; done in order it equals `forward`."""
    from graphvae import _hip
    from graphvae.engine import PHASE_ASSIGN_LOSS, PHASE_MATCH
    G = load_g5(golden_dir / "g5_forward_rand_deg.npz")
    model, batch, eps = _model_and_batch(G, "stream")
    model.train()
    eng = model._engine_for(batch)
    args = (model.flat_parameters(), model.bn_state().clone(), batch, eps, model._pos_weight_dev(), True)
    with pytest.raises(_hip.GvaeHipError):
        eng.forward_match_loss()
    ref = eng.forward(*args).clone()
    perm_ref = eng.perm.clone()
    args = (model.flat_parameters(), model.bn_state().clone(), batch, eps, model._pos_weight_dev(), True)
    eng.forward_model(*args)
    with pytest.raises(_hip.GvaeHipError):
        eng.forward_match_loss(PHASE_ASSIGN_LOSS)
    eng.forward_match_loss(PHASE_MATCH)
    with pytest.raises(_hip.GvaeHipError):
        eng.forward_match_loss(PHASE_MATCH)
    out = eng.forward_match_loss(PHASE_ASSIGN_LOSS).clone()
    with pytest.raises(_hip.GvaeHipError):
        eng.forward_match_loss(PHASE_ASSIGN_LOSS)
    assert torch.equal(out, ref) and torch.equal(eng.perm, perm_ref)
    # the matching in its three parts: no part before its predecessor, all four in order = forward
    from graphvae.engine import PHASE_MPM, PHASE_PREP, PHASE_TAIL
    args = (model.flat_parameters(), model.bn_state().clone(), batch, eps, model._pos_weight_dev(), True)
    eng.forward_model(*args)
    for early in (PHASE_MPM, PHASE_TAIL, PHASE_PREP | PHASE_TAIL, PHASE_MPM | PHASE_TAIL):
        with pytest.raises(_hip.GvaeHipError):
            eng.forward_match_loss(early)
    eng.forward_match_loss(PHASE_PREP)
    with pytest.raises(_hip.GvaeHipError):
        eng.forward_match_loss(PHASE_TAIL)
    eng.forward_match_loss(PHASE_MPM)
    eng.forward_match_loss(PHASE_TAIL)
    out = eng.forward_match_loss(PHASE_ASSIGN_LOSS).clone()
    assert torch.equal(out, ref) and torch.equal(eng.perm, perm_ref)


@pytest.mark.parametrize("matching", ["fused", "stream"])
def test_forward_old_vs_reference_golden(golden_dir, matching):
    """The legacy loss (reference model.py:378-440) on a batch with padded nodes: losses,
    permutations and every parameter gradient against the imported reference's."""
    G, L = load_g10(golden_dir)
    model, batch, eps = _model_and_batch(G, matching)
    model.train()
    out = model.forward_with_eps(batch, eps, legacy=True)
    assert "loss" not in out
    for k in ("rec", "kl", "max_logit", "logvar_max"):
        assert rel_err(out[k].item(), L.out[k]) < TOL, (k, out[k].item(), L.out[k])
    B = batch.num_graphs
    assert rel_err(model._engine.debug_tensor("rec_i", B).cpu().numpy(), L.out["rec_i"]) < TOL
    X = model._engine.debug_tensor("X", B).cpu().numpy()
    perm = model.last_permutations().cpu().numpy()
    for g, p in enumerate(L.perms):
        assert perm_equivalent(perm[g, :len(p)], p, X[g])
    (out["rec"] + out["kl"]).backward()
    got = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    check_grads_accurate(got, L.grads, truth64(G, legacy_mask=True), TOL)
    torch.manual_seed(0)
    out2 = model.forward_old(batch)
    assert set(out2) == {"rec", "kl", "max_logit", "logvar_max"}


@pytest.mark.parametrize("matching", ["fused", "stream"])
def test_individual_rho_vs_reference_golden(golden_dir, matching):
    """`forward_individual_rho` (reference model_legacy.py:257-363): the ground truth is permuted and
    the prediction is not, so a wrong permutation would change the loss.  Losses, the matching over
    all max_nodes rows, permutations and every gradient against the imported reference's."""
    from oracle import matching as ora
    G, L = load_g12(golden_dir)
    model, batch, eps = _model_and_batch(G, matching)
    model.rho_global = None                      # not needed by this variant
    model.train()
    out = model.forward_with_eps(batch, eps, individual_rho=True)
    assert "loss" not in out
    for k in ("rec", "kl", "max_logit", "logvar_max"):
        assert rel_err(out[k].item(), L.out[k]) < TOL, (k, out[k].item(), L.out[k])
    B, N = batch.num_graphs, batch.max_nodes
    assert rel_err(model._engine.debug_tensor("rec_i", B).cpu().numpy(), L.out["rec_i"]) < TOL
    assert model.last_mpm_iterations().min() >= 1
    X = model._engine.debug_tensor("X", B).cpu().numpy()
    P = model._engine.debug_tensor("P", B).cpu().numpy()
    perm = model.last_permutations().cpu().numpy()
    A = batch.adj_dense.cpu().numpy()
    for g, p in enumerate(L.perms):
        off = ~np.eye(N, dtype=bool)
        assert rel_err(P[g][off], L.P[g][off]) < TOL           # the diagonal is read as 1 by the matching
        assert perm_equivalent(perm[g], p, X[g])
        # the device's own (A, P) through the C oracle: X bit for bit, permutation exactly
        Ag = A[g].copy()
        np.fill_diagonal(Ag, 1.0)
        Xo, _ = ora.mpm_sfree(Ag, P[g])
        assert np.array_equal(X[g], Xo)
        assert np.array_equal(perm[g], ora.assign_perm(Xo))
    (out["rec"] + out["kl"]).backward()
    got = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    want = dict(L.grads)
    for n in got:
        if n.startswith("dec_node"):
            assert not got[n].any()
            want[n] = got[n]
    check_grads_accurate(got, want, truth64(G, individual_rho=True), TOL)
    torch.manual_seed(0)
    out2 = model.forward_individual_rho(batch)
    assert set(out2) == {"rec", "kl", "max_logit", "logvar_max"}
    # the identity permutation gives another loss: this variant consumes the matching
    G5 = load_g5(golden_dir / "g5_forward_rand_id.npz")
    assert abs(out["rec"].item() - float(G5.out["rec"])) > 0.1


def test_autograd_path_equals_fused_step_path(golden_dir):
    """model(data) + loss.backward() and the trainer's flat-buffer path share kernels:
    same gradients, bit for bit."""
    G = load_g5(golden_dir / "g5_forward_rand_deg.npz")
    model, batch, eps = _model_and_batch(G)
    model.train()
    out = model.forward_with_eps(batch, eps)
    (out["rec"] + 0.25 * out["kl"]).backward()
    g_auto = torch.cat([p.grad.reshape(-1) for p in model._named_param_list()])
    eng = model._engine
    flat_g = torch.zeros_like(model.flat_parameters())
    eng.forward(model.flat_parameters(), model.bn_state().clone(), batch, eps, model._pos_weight_dev(), True)
    eng.backward(model.flat_parameters(), model._pos_weight_dev(), flat_g, 1.0, 0.25)
    g_flat = torch.cat([v.reshape(-1) for v in model._views_of(flat_g)])
    assert torch.equal(g_auto, g_flat)


@pytest.mark.parametrize("feature_type,N,B,kind", [("deg", 32, 8, "dense"), ("id", 24, 6, "enzymes"),
                                                    ("deg", 40, 6, "enzymes")])
def test_default_sizes_vs_cpu_restatement(feature_type, N, B, kind):
    """hid 128 / z 64 (trainer defaults) on synthetic graphs: loss, per-graph terms and all
    gradients against the CPU restatement on identical weights and noise."""
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    hb = synthetic_batch(kind, B, N, feature_type, seed=5)
    F = hb.x.shape[-1]
    torch.manual_seed(7)
    model = GraphVAE(F, 128, 64, N)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("bias") or n_.startswith("bn"):
                p.add_(0.1 * torch.randn(p.shape))
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.precompute_dataset_rho([hb])
    state0["pos_weight_global"] = model.pos_weight_global.detach().clone()
    eps = torch.randn(B, 64)
    model = model.to(DEV).train()
    out = model.forward_with_eps(hb.to(DEV), eps.to(DEV))
    out["loss"].backward()

    st = R.make_trainable(state0)
    graphs = R.graphs_from_arrays(hb.x, hb.adj_dense, hb.n_x, hb.num_real_nodes)
    res = R.forward_batch(st, graphs, [eps[g:g + 1] for g in range(B)], train=True)
    res.loss.backward()
    for k in ("rec", "kl", "loss", "max_logit", "logvar_max"):
        assert rel_err(out[k].item(), getattr(res, k).item()) < TOL, k
    perm = model.last_permutations().cpu().numpy()
    X = model._engine.debug_tensor("X", B).cpu().numpy()
    n_equal = 0
    for g, o in enumerate(res.per_graph):
        r = hb.r_host[g]
        assert perm_equivalent(perm[g, :r], o.perm[:r].numpy(), X[g])
        n_equal += np.array_equal(perm[g], o.perm.numpy())
    assert n_equal >= B - 1
    ref_g = {n: st[n].grad.numpy() for n in R.PARAM_NAMES}
    got = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    # float64 yardstick for the CPU restatement's own rounding
    G = type("G", (), {})()
    G.state0, G.graphs, G.eps = state0, graphs, [eps[g:g + 1] for g in range(B)]
    check_grads_accurate(got, ref_g, truth64(G), TOL)


def test_chunked_streamed_matching_equals_one_chunk():
    """A similarity-tensor budget smaller than the batch's tensor makes the step walk the
    batch in chunks through one S buffer: same losses, permutations and gradients, bit for
    bit, as with everything resident."""
    from graphvae import _hip
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    N, B = 24, 12
    hb = synthetic_batch("enzymes", B, N, "deg", seed=11)
    eps = torch.randn(B, 16, generator=torch.Generator().manual_seed(3))
    results = []
    for budget in (None, 3 * _hip.s_block_floats(N, _hip.S_PADDED) * 4):
        torch.manual_seed(4)
        model = GraphVAE(1, 32, 16, N, matching="stream", s_budget_bytes=budget)
        model.precompute_dataset_rho([hb])
        model = model.to(DEV).train()
        # a workspace full of non-zero words: nothing in it may be read before it is written
        # (recycled allocator blocks are not zero)
        dev_batch = hb.to(DEV)
        eng = model._engine_for(dev_batch)
        eng._ensure(B)
        eng._ws.fill_(0xA5)
        out = model.forward_with_eps(dev_batch, eps.to(DEV))
        out["loss"].backward()
        n_chunks = len(model._engine._S_chunks) - 1
        grads = torch.cat([p.grad.reshape(-1) for p in model._named_param_list()])
        results.append((n_chunks, out["rec"].item(), out["kl"].item(),
                        model.last_permutations().clone(), model._engine.iters_done.clone(), grads))
    (n1, rec1, kl1, perm1, it1, g1), (n2, rec2, kl2, perm2, it2, g2) = results
    assert n1 == 1 and n2 >= 3
    assert rec1 == rec2 and kl1 == kl2
    assert torch.equal(perm1, perm2) and torch.equal(it1, it2) and torch.equal(g1, g2)


def test_forward_requires_rho_and_gpu():
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    from graphvae._hip import GvaeHipError
    hb = synthetic_batch("dense", 2, 8, "deg")
    model = GraphVAE(1, 8, 4, 8)
    with pytest.raises(RuntimeError):
        model(hb)
    model.rho_global = 0.5
    with pytest.raises(GvaeHipError):
        model(hb)                       # CPU model + CPU batch: no fallback


def test_backward_accumulates_over_micro_batches(golden_dir):
    """Two forward/backward passes without zeroing: .grad holds g1 + g2 (the autograd bridge
    hands out fresh tensors; views of one persistent buffer would give 2 * g2)."""
    G = load_g5(golden_dir / "g5_forward_rand_deg.npz")
    model, batch, eps = _model_and_batch(G)
    model.train()
    singles = []
    for scale in (1.0, 3.0):
        model.zero_grad(set_to_none=True)
        out = model.forward_with_eps(batch, eps)
        (scale * out["loss"]).backward()
        singles.append([p.grad.detach().clone() for p in model._named_param_list()])
    model.zero_grad(set_to_none=True)
    for scale in (1.0, 3.0):
        out = model.forward_with_eps(batch, eps)
        (scale * out["loss"]).backward()
    for p, g1, g2 in zip(model._named_param_list(), *singles):
        assert torch.equal(p.grad, g1 + g2)
    # zero_grad(set_to_none=False) keeps the tensors: the next backward must not see stale values
    model.zero_grad(set_to_none=False)
    out = model.forward_with_eps(batch, eps)
    out["loss"].backward()
    for p, g1 in zip(model._named_param_list(), singles[0]):
        assert torch.equal(p.grad, g1)


def test_eval_mode_backward_vs_cpu_restatement():
    """After model.eval() the forward normalises with the running statistics, which are
    constants of the input: the BatchNorm backward is then dY = dpre * invstd * gamma (no batch
    mean / variance terms).  Gradients against autograd of the CPU restatement in eval mode."""
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    N, B = 16, 6
    hb = synthetic_batch("enzymes", B, N, "deg", seed=8)
    torch.manual_seed(12)
    model = GraphVAE(1, 32, 16, N)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("bias") or n_.startswith("bn"):
                p.add_(0.1 * torch.randn(p.shape))
        for bn in (model.bn1, model.bn2):                 # non-trivial running statistics
            bn.running_mean.add_(0.2 * torch.randn(32))
            bn.running_var.mul_(1.0 + 0.5 * torch.rand(32))
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.precompute_dataset_rho([hb])
    state0["pos_weight_global"] = model.pos_weight_global.detach().clone()
    eps = torch.randn(B, 16)
    model = model.to(DEV).eval()
    out = model.forward_with_eps(hb.to(DEV), eps.to(DEV))
    out["loss"].backward()
    st = R.make_trainable(state0)
    graphs = R.graphs_from_arrays(hb.x, hb.adj_dense, hb.n_x, hb.num_real_nodes)
    res = R.forward_batch(st, graphs, [eps[g:g + 1] for g in range(B)], train=False)
    res.loss.backward()
    for k in ("rec", "kl", "loss"):
        assert rel_err(out[k].item(), getattr(res, k).item()) < TOL, k
    for n_, p in model.named_parameters():
        ref = st[n_].grad.numpy()
        got = p.grad.detach().cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-6), n_
    # and the training-mode gradients of the same batch differ (the flag is not a no-op)
    model.train()
    model.zero_grad(set_to_none=True)
    out = model.forward_with_eps(hb.to(DEV), eps.to(DEV))
    out["loss"].backward()
    g_train = model.conv1.lin.weight.grad.cpu().numpy()
    assert np.abs(g_train - st["conv1.lin.weight"].grad.numpy()).max() > 1e-3 * np.abs(g_train).max()


@pytest.mark.gpu
@pytest.mark.parametrize("N,train", [(32, True), (40, True), (24, False), (100, True)])
def test_model_half_resident_kernel_equals_generic_kernel(N, train):
    """The model half of the forward has a kernel that keeps a graph's activations in LDS (graphs
    of at most 128 nodes, hidden width 128) and a generic one that stages them through global
    memory; the first reads weight rows as 16-byte vectors, so a parameter buffer that starts
    4 bytes off that alignment takes the generic one.  Same activations, logits and BatchNorm
    statistics from both: the encoder of either runs in fp64 (matrix cores in one, plain fma chains
    in the other), so the fp32 tensors they leave agree to a few units in the last place."""
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    B = 7
    hb = synthetic_batch("enzymes", B, N, "deg", seed=21).to(DEV)
    torch.manual_seed(5)
    model = GraphVAE(1, 128, 64, N).to(DEV)
    model.precompute_dataset_rho([hb])
    model.train(train)
    eng = model._engine_for(hb)
    eps = torch.randn(B, 64, device=DEV)
    flat = model.flat_parameters()
    shifted = torch.empty(flat.numel() + 1, dtype=torch.float32, device=DEV)[1:]
    shifted.copy_(flat)
    assert flat.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    names = ("Ahat", "Y1", "H1", "Y2", "H2", "gvec", "mu", "logvar_raw", "vec_logits", "node_logits")
    got = []
    for params in (flat, shifted):
        bn = model.bn_state().clone()
        eng.forward_model(params, bn, hb, eps, model._pos_weight_dev(), train)
        got.append({k: eng.debug_tensor(k, B) for k in names} | {"bn_state": bn})
        eng.forward_match_loss()
    for k in got[0]:
        a, b = got[0][k], got[1][k]
        scale = max(b.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= 2e-6 * scale, (k, (a - b).abs().max().item(), scale)


@pytest.mark.gpu
def test_accumulate_loss_entry_point():
    """gvae_step_accumulate_loss: the float64 accumulator takes rec + beta * kl evaluated in fp32
    (two roundings, as `out4[0] + beta * out4[1]` of the step-by-step path), the copy takes the
    four scalars."""
    from graphvae.engine import StepEngine
    eng = StepEngine(1, 8, 4, 6, DEV)
    out4 = torch.tensor([3.1415927, 0.57721566, 7.0, -2.0], dtype=torch.float32, device=DEV)
    acc = torch.tensor([1.5], dtype=torch.float64, device=DEV)
    copy = torch.zeros(4, dtype=torch.float32, device=DEV)
    beta = 0.3
    for _ in range(2):
        eng.accumulate_loss(out4, beta, acc, copy=copy)
    step = (out4[0] + beta * out4[1]).double().item()
    assert acc.item() == 1.5 + step + step
    assert torch.equal(copy, out4)
    eng.accumulate_loss(out4, beta, acc)                  # the copy is optional
    assert acc.item() == 1.5 + step + step + step


@pytest.mark.gpu
def test_step_refuses_high_priority_streams():
    """TrainStep with its matching on side streams refuses to run when the caller's current
    stream is a high-priority one (the guard bench.py applies too): the error comes before any
    kernel of the step is enqueued."""
    from graphvae import _hip
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    from graphvae.train.train_ddp import TrainStep
    N, B = 16, 8
    hb = synthetic_batch("enzymes", B, N, "deg", seed=5)
    torch.manual_seed(1)
    model = GraphVAE(1, 32, 16, N).to(DEV)
    model.precompute_dataset_rho([hb])
    model.train()
    step = TrainStep(model, 1e-3, 0.0, 5.0, pipeline=True)
    high = torch.cuda.Stream(device=DEV, priority=-1)
    with torch.cuda.stream(high):
        with pytest.raises(_hip.GvaeHipError, match="default priority"):
            step(hb.to(DEV), 0.5, 1.0)
    torch.cuda.synchronize()
    out = step(hb.to(DEV), 0.5, 1.0)          # the same object runs from a default-priority stream
    step.wait()
    assert torch.isfinite(out).all()
