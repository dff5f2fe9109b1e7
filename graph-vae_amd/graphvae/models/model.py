"""GraphVAE with the reference's model API, running on hand-written HIP kernels.

Drop-in surface (reference graphvae/models/model.py):
  GraphVAE(in_dim, hid_dim, z_dim, max_nodes, *, pool="sum")          :63-94
  .precompute_dataset_rho(loader)                                       :99-118
  .forward(data) -> {"rec","kl","loss","max_logit","logvar_max"}        :281-373
  ._edge_sim_tensor(A, B) -> (R,R,R,R)                                  :171-216
  ._mpm(X0, S, iters=50, tol=1e-4, early_stop=True) -> (R,R)            :247-276
  ._reparam(mu, logvar), vec_to_adj(vec, N, diag=False), deg_sim(d1, d2)
  .sample_decode(z, th_edge, th_node): the sampler's decoder loop as one call  (experiments/sample.py:50-73)
State-dict keys are the reference's (conv{1,2}.{bias,lin.weight}, bn{1,2}.*, mu.*,
logvar.*, dec_edge.{0,2}.*, dec_node.{0,2}.*, pos_weight_global), so checkpoints move
between the two implementations.

All parameters live in one flat fp32 device buffer (the module's Parameters are views
into it); the whole batch goes through libgvae_hip.so in a handful of C calls instead
of the reference's per-graph Python loop.  Gradients come back as views of one flat
buffer, ready for a single RCCL all-reduce.  There is no CPU path: `forward` raises if
the model or the data is not on a GPU.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..data import DenseGraphBatch, as_dense_batch
from ..engine import (LOSS_CURRENT, LOSS_DUMMY_MASKED, LOSS_INDIVIDUAL_RHO, PARAM_NAMES, StepEngine,
                      param_layout)


def vec_to_adj(vec: torch.Tensor, N: int, *, diag: bool = False) -> torch.Tensor:
    """Upper-triangular off-diagonal vector (row-major pair order) -> symmetric (N,N);
    also accepts a leading batch dimension.  Reference model.py:42-51."""
    iu = torch.triu_indices(N, N, offset=1, device=vec.device)
    out = vec.new_zeros(vec.shape[:-1] + (N, N))
    out[..., iu[0], iu[1]] = vec
    out = out + out.transpose(-1, -2)
    if diag:
        d = torch.arange(N, device=vec.device)
        out[..., d, d] = 1.0
    return out


def deg_sim(d1, d2):
    """1 / (|d1 - d2| + 1).  Reference model.py:54-56."""
    return (torch.abs(d1 - d2) + 1.0).reciprocal()


class _WeightOnly(nn.Module):
    """`lin` of a GCN layer: a weight matrix without bias (state-dict key `lin.weight`)."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim))

    def forward(self, x):
        return x @ self.weight.t()


class DenseGCNConv(nn.Module):
    """Parameter holder with PyG GCNConv's state-dict layout (`bias`, `lin.weight`).
    The training step never calls this module; `forward` is a dense convenience for
    inspecting a layer outside the step (self loop on every row, D^-1/2 (A+I) D^-1/2)."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.lin = _WeightOnly(in_dim, out_dim)

    def forward(self, x, adj):
        a = (adj != 0).to(x.dtype)
        eye = torch.eye(a.shape[-1], dtype=x.dtype, device=x.device)
        a = a * (1 - eye) + eye
        dinv = a.sum(-1).rsqrt()
        return (dinv.unsqueeze(-1) * a * dinv.unsqueeze(-2)) @ self.lin(x) + self.bias


def _mpm_run(X0, S, iters, tol, early_stop, dev):
    """(X, iterations executed) of the streamed MPM kernel on one dense S."""
    R = X0.shape[0]
    X = X0.detach().to(dev, torch.float32).clone().contiguous().unsqueeze(0)
    Sd = S.detach().to(dev, torch.float32).contiguous().view(-1)
    Rt = torch.tensor([R], dtype=torch.int32, device=dev)
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    done = _hip.mpm_stream(Sd, off, Rt, R, X, iters, tol, early_stop)
    return X[0], done


class _MpmFunction(torch.autograd.Function):
    """Autograd of `_mpm`.  Forward: the streamed kernel.  Backward: the trajectory is recomputed
    one iteration per launch (same kernel, hence the same bits), then walked backwards; every
    step differentiates the normalisation and replays the maxima of the pooling
    (gvae_mpm_pool_bwd).  Limits: one graph of at most 255 nodes with a dense (R,R,R,R) gradient
    w.r.t. S; every X_t of the trajectory is kept.  The pooling backward adds the shares of
    X[j,b] with float atomics, so the gradient (unlike every forward result of the library) is
    not bit-reproducible from run to run; the reference pins only that it is finite
    (model_legacy2.py:670-684).  Not on the training path: the step never differentiates the
    matching (model.py:319 detaches X)."""

    @staticmethod
    def forward(ctx, X0, S, iters, tol, early_stop, dev):
        X, done = _mpm_run(X0, S, iters, tol, early_stop, dev)
        ctx.save_for_backward(X0, S)
        ctx.args = (int(done[0]), tol, dev)
        return X.to(X0.device)

    @staticmethod
    def backward(ctx, gX):
        X0, S = ctx.saved_tensors
        done, tol, dev = ctx.args
        Sd = S.detach().to(dev, torch.float32).contiguous()
        traj = [X0.detach().to(dev, torch.float32).contiguous()]
        for _ in range(done):
            traj.append(_mpm_run(traj[-1], Sd, 1, tol, False, dev)[0])
        g = gX.detach().to(dev, torch.float32).contiguous()
        gS = torch.zeros_like(Sd) if ctx.needs_input_grad[1] else None
        for t in range(done - 1, -1, -1):
            Xt, Xn = traj[t], traj[t + 1]
            nrm = _hip.mpm_pool_fwd(Sd, Xt).norm(p=2)
            gY = ((g - Xn * (g * Xn).sum()) / nrm).contiguous()
            g = _hip.mpm_pool_bwd(Sd, Xt, gY, gS)
        return (g.to(X0.device) if ctx.needs_input_grad[0] else None,
                gS.to(S.device) if gS is not None else None, None, None, None, None)


class _StepFunction(torch.autograd.Function):
    """Autograd bridge: forward and backward both run inside libgvae_hip.so."""

    @staticmethod
    def forward(ctx, model, batch, eps, loss_variant, *params):
        eng = model._engine_for(batch)
        out4 = eng.forward(model._flat, model._bn_state, batch, eps, model._pos_weight_dev(),
                           model.training, loss_variant).clone()
        ctx.model, ctx.engine, ctx.generation = model, eng, eng.generation
        rec, kl, max_logit, logvar_max = out4[0], out4[1], out4[2], out4[3]
        ctx.mark_non_differentiable(max_logit, logvar_max)
        return rec, kl, max_logit, logvar_max

    @staticmethod
    def backward(ctx, d_rec, d_kl, _d_ml, _d_lv):
        model, eng = ctx.model, ctx.engine
        if eng.generation != ctx.generation:
            raise RuntimeError("GraphVAE: backward() after another forward(); the step keeps the "
                               "activations of the most recent forward only")
        up = torch.stack([d_rec if d_rec is not None else torch.zeros_like(eng.out4[0]),
                          d_kl if d_kl is not None else torch.zeros_like(eng.out4[0])]).contiguous()
        # a fresh buffer per call: autograd keeps the returned tensors as .grad without copying,
        # so views of a persistent buffer would alias the next backward's output (and
        # accumulation over micro-batches would add a gradient onto itself)
        grads = torch.zeros_like(model._flat)
        eng.backward(model._flat, model._pos_weight_dev(), grads, 1.0, 1.0, up)
        views = model._views_of(grads)
        return (None, None, None, None) + tuple(views)


class GraphVAE(nn.Module):
    def __init__(self, in_dim: int, hid_dim: int, z_dim: int, max_nodes: int, *, pool: str = "sum",
                 matching: str = "auto", mpm_iters: int = 50, mpm_tol: float = 1e-4,
                 mpm_early_stop: bool = True, s_budget_bytes: int | None = None) -> None:
        super().__init__()
        # bytes of HBM the streamed matching may use for the similarity tensor at a time
        # (None: a third of the device); larger batches are matched in chunks
        self.s_budget_bytes = s_budget_bytes
        self.in_dim, self.hid_dim, self.z_dim = int(in_dim), int(hid_dim), int(z_dim)
        self.max_nodes = int(max_nodes)
        self.pool = pool
        self.matching = matching
        self.mpm_iters, self.mpm_tol, self.mpm_early_stop = mpm_iters, mpm_tol, mpm_early_stop

        pairs = self.max_nodes * (self.max_nodes - 1) // 2
        self.conv1 = DenseGCNConv(in_dim, hid_dim)
        self.bn1 = nn.BatchNorm1d(hid_dim)
        self.conv2 = DenseGCNConv(hid_dim, hid_dim)
        self.bn2 = nn.BatchNorm1d(hid_dim)
        self.mu = nn.Linear(hid_dim, z_dim)
        self.logvar = nn.Linear(hid_dim, z_dim)
        self.dec_edge = nn.Sequential(nn.Linear(z_dim, hid_dim), nn.ReLU(), nn.Linear(hid_dim, pairs))
        self.dec_node = nn.Sequential(nn.Linear(z_dim, hid_dim), nn.ReLU(), nn.Linear(hid_dim, max_nodes))
        self.register_buffer("pos_weight_global", torch.tensor(1.0))
        self.rho_global: float | None = None

        self._init_weights()
        self._offsets, self._sizes, self._flat_len = None, None, None
        self._flat = None
        self._bn_state = None
        self._engine = None
        self._flatten()

    # ------------------------------------------------------------------ parameters
    def _init_weights(self) -> None:
        """Xavier-uniform with ReLU gain on every weight matrix, zero biases, BatchNorm
        gamma 1 / beta 0 (reference model.py:123-134)."""
        gain = math.sqrt(2.0)
        for name, p in self.named_parameters():
            if name.startswith("bn"):
                (nn.init.ones_ if name.endswith("weight") else nn.init.zeros_)(p)
            elif p.dim() == 2:
                nn.init.xavier_uniform_(p, gain=gain)
            else:
                nn.init.zeros_(p)

    def _named_param_list(self):
        byname = dict(self.named_parameters())
        return [byname[n] for n in PARAM_NAMES]

    def _layout(self):
        if self._offsets is None:
            if _hip.lib_path().exists():
                self._offsets, self._sizes, self._flat_len = param_layout(
                    self.in_dim, self.hid_dim, self.z_dim, self.max_nodes)
            else:  # lets the module be built (e.g. to inspect a checkpoint) without the library
                sizes = [p.numel() for p in self._named_param_list()]
                offs, tot = [], 0
                for s in sizes:
                    offs.append(tot)
                    tot += (s + 3) // 4 * 4
                self._offsets, self._sizes, self._flat_len = offs, sizes, tot
        return self._offsets, self._sizes, self._flat_len

    def _views_of(self, flat: torch.Tensor):
        offs, sizes, _ = self._layout()
        return [flat[o:o + s].view(p.shape) for o, s, p in zip(offs, sizes, self._named_param_list())]

    def _flatten(self) -> None:
        """(Re)build the flat parameter / BatchNorm-state buffers on the parameters' current
        device and make every Parameter and running statistic a view of them."""
        offs, sizes, tot = self._layout()
        params = self._named_param_list()
        dev = params[0].device
        flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        for view, p in zip(self._views_of(flat), params):
            view.copy_(p.data)
            p.data = view
        H = self.hid_dim
        bn = torch.empty(4 * H, dtype=torch.float32, device=dev)
        for k, (mod, key) in enumerate(((self.bn1, "running_mean"), (self.bn1, "running_var"),
                                        (self.bn2, "running_mean"), (self.bn2, "running_var"))):
            view = bn[k * H:(k + 1) * H]
            view.copy_(mod._buffers[key])
            mod._buffers[key] = view
        self._flat, self._bn_state = flat, bn
        self._engine = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def flat_parameters(self) -> torch.Tensor:
        """The single fp32 buffer all parameters are views of."""
        return self._flat

    def bn_state(self) -> torch.Tensor:
        """[bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var] flat."""
        return self._bn_state

    def _pos_weight_dev(self) -> torch.Tensor:
        return self.pos_weight_global.reshape(1)

    def _engine_for(self, batch: DenseGraphBatch) -> StepEngine:
        dev = self._flat.device
        if dev.type != "cuda":
            raise _hip.GvaeHipError("GraphVAE runs on the GPU only: move the model with .to('cuda')")
        if batch.device != dev:
            raise _hip.GvaeHipError(f"batch on {batch.device}, model on {dev}")
        if self._engine is None:
            self._engine = StepEngine(self.in_dim, self.hid_dim, self.z_dim, self.max_nodes, dev,
                                      matching=self.matching, mpm_iters=self.mpm_iters,
                                      tol=self.mpm_tol, early_stop=self.mpm_early_stop,
                                      s_budget_bytes=self.s_budget_bytes)
        return self._engine

    # ------------------------------------------------------------------ dataset statistics
    @torch.no_grad()
    def precompute_dataset_rho(self, loader) -> None:
        """Edge density over the upper triangles of all real-node blocks and the clipped
        positive weight (1-rho)/(rho+1e-8) in [1, 20].  Reference model.py:99-118."""
        pos_edges, total_pairs = 0.0, 0.0
        for data in loader:
            b = as_dense_batch(data, self.max_nodes)
            N = b.max_nodes
            idx = torch.arange(N, device=b.device)
            real = idx.view(1, N) < b.num_real_nodes.view(-1, 1)
            mask = real.unsqueeze(2) & real.unsqueeze(1) & (idx.view(1, N, 1) < idx.view(1, 1, N))
            pos_edges += float((b.adj_dense * mask).sum(dtype=torch.float64))
            total_pairs += sum(r * (r - 1) / 2 for r in b.r_host)
        rho = pos_edges / (total_pairs + 1e-8)
        w = float(np.clip((1 - rho) / (rho + 1e-8), 1.0, 20.0))
        self.rho_global = rho
        self.pos_weight_global.fill_(w)
        print(f"[GraphVAE] rho(dataset) = {rho:.4f}  ->  pos_weight = {w:.3f}")

    # ------------------------------------------------------------------ helpers of the API
    @staticmethod
    def _reparam(mu, logvar):
        """mu + eps * exp(logvar / 2).  Reference model.py:137-139."""
        return mu + torch.randn_like(logvar) * torch.exp(0.5 * logvar)

    def _hip_device(self, *tensors) -> torch.device:
        for t in tensors:
            if t.is_cuda:
                return t.device
        if self._flat.is_cuda:
            return self._flat.device
        if torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        raise _hip.GvaeHipError("the matching kernels need a GPU (no CPU fallback)")

    @torch.no_grad()
    def _edge_sim_tensor(self, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """(R,R,R,R) edge-similarity tensor, R = number of rows of A with positive sum.
        Same values, bit for bit, as reference model.py:171-216; returned on A's device."""
        dev = self._hip_device(A, B)
        Ad = A.detach().to(dev, torch.float32)
        Bd = B.detach().to(dev, torch.float32)
        R = int((Ad.sum(1) > 0).sum())
        if R == 0:
            return A.new_zeros(0, 0, 0, 0, dtype=torch.float32)
        Ad = Ad[:R, :R].contiguous().unsqueeze(0)
        Bd = Bd[:R, :R].contiguous().unsqueeze(0)
        Rt = torch.tensor([R], dtype=torch.int32, device=dev)
        NS = _hip.node_sim(Ad, Bd, Rt)
        S, _ = _hip.edge_sim(Ad, Bd, NS, Rt, [R])
        return S[: R ** 4].view(R, R, R, R).to(A.device)

    def _mpm(self, X0: torch.Tensor, S: torch.Tensor, iters: int = 50, tol: float = 1e-4,
             early_stop: bool = True) -> torch.Tensor:
        """Max-pool matching on an explicit S (any non-negative tensor), bit-identical to
        reference model.py:247-276, and differentiable w.r.t. X0 and S like it (the reference
        checks the gradient w.r.t. X0, model_legacy2.py:670-684; `forward` uses the result
        detached, model.py:319)."""
        R = X0.shape[0]
        if tuple(S.shape) != (R, R, R, R):
            raise ValueError(f"S must be ({R},{R},{R},{R}), got {tuple(S.shape)}")
        dev = self._hip_device(X0, S)
        if torch.is_grad_enabled() and (X0.requires_grad or S.requires_grad):
            return _MpmFunction.apply(X0, S, int(iters), float(tol), bool(early_stop), dev)
        return _mpm_run(X0, S, iters, tol, early_stop, dev)[0].to(X0.device)

    def _mpm_loop(self, X0: torch.Tensor, S: torch.Tensor, iters: int = 50) -> torch.Tensor:
        """The reference's loop form (model.py:218-245): `iters` iterations over all max_nodes
        rows without early stop.  Same kernel as `_mpm`; the two agree bit for bit here (the
        reference's two forms agree to rounding, model_legacy2.py:649-667)."""
        if tuple(X0.shape) != (self.max_nodes, self.max_nodes):
            raise ValueError(f"X0 must be ({self.max_nodes},{self.max_nodes})")
        return self._mpm(X0, S, iters=iters, early_stop=False)

    @torch.no_grad()
    def edge_sim_tensor_loop(self, A: torch.Tensor, B: torch.Tensor, degA: torch.Tensor,
                             degB: torch.Tensor) -> torch.Tensor:
        """The reference's loop form of the similarity tensor (model.py:152-169) over all
        max_nodes rows, with the diagonals of A and B as given (no self-loop correction) and the
        degrees supplied by the caller: S[i,i,a,a] = A[i,i] B[a,a] / (|degA[i] - degB[a]| + 1),
        S[i,j,a,b] = A[i,j] A[i,i] A[j,j] B[a,b] B[a,a] B[b,b] for i != j and a != b, else 0.
        The diagonal factors are folded into the two matrices before the tensor kernel forms the
        products, so entries equal the loop's up to the rounding of that regrouping (equal bit
        for bit when the diagonals are 0 or 1)."""
        N = self.max_nodes
        dev = self._hip_device(A, B)
        Ad = A.detach().to(dev, torch.float32)[:N, :N]
        Bd = B.detach().to(dev, torch.float32)[:N, :N]
        dA, dB = Ad.diagonal().clone(), Bd.diagonal().clone()
        node = dA.view(N, 1) * dB.view(1, N) / ((degA.to(dev, torch.float32).view(N, 1)
                                                 - degB.to(dev, torch.float32).view(1, N)).abs() + 1.0)
        As = (Ad * dA.view(N, 1) * dA.view(1, N)).contiguous().unsqueeze(0)
        Bs = (Bd * dB.view(N, 1) * dB.view(1, N)).contiguous().unsqueeze(0)
        Rt = torch.tensor([N], dtype=torch.int32, device=dev)
        S, _ = _hip.edge_sim(As, Bs, node.contiguous().unsqueeze(0), Rt, [N])
        return S[: N ** 4].view(N, N, N, N).to(A.device)

    # ------------------------------------------------------------------ forward
    def forward(self, data):
        """Training / validation forward of a whole batch.  `data`: DenseGraphBatch, a list
        of graph records, an object with to_data_list() or one graph record, on the model's
        GPU.  Returns 0-dim tensors; rec / kl / loss are differentiable w.r.t. the parameters."""
        if self.rho_global is None:
            raise RuntimeError("Call `model.precompute_dataset_rho(...)` first!")
        batch = as_dense_batch(data, self.max_nodes)
        eps = torch.randn(batch.num_graphs, self.z_dim, dtype=torch.float32, device=batch.device)
        return self.forward_with_eps(batch, eps)

    def forward_with_eps(self, batch: DenseGraphBatch, eps: torch.Tensor, *,
                         legacy: bool = False, individual_rho: bool = False) -> dict:
        """`forward` (or, with legacy=True, `forward_old`; with individual_rho=True,
        `forward_individual_rho`) with caller-supplied reparameterisation noise (one row per
        graph)."""
        if self.rho_global is None and not individual_rho:
            raise RuntimeError("Call `model.precompute_dataset_rho(...)` first!")
        variant = LOSS_INDIVIDUAL_RHO if individual_rho else (LOSS_DUMMY_MASKED if legacy else LOSS_CURRENT)
        rec, kl, max_logit, logvar_max = _StepFunction.apply(self, batch, eps, variant,
                                                             *self._named_param_list())
        if self.training:
            for bn in (self.bn1, self.bn2):
                bn.num_batches_tracked += batch.num_graphs
        if legacy or individual_rho:   # reference model.py:439-440 / model_legacy.py:363: no "loss" entry
            return {"rec": rec, "kl": kl, "max_logit": max_logit, "logvar_max": logvar_max}
        return {"rec": rec, "kl": kl, "loss": rec + kl, "max_logit": max_logit,
                "logvar_max": logvar_max}

    def forward_old(self, data):
        """The older loss kept by the reference (model.py:378-440): as `forward`, except that the
        node logits of padded entries are overwritten with -10 before the node loss, so padded
        entries add a constant and get no gradient.  Same kernels, same matching."""
        if self.rho_global is None:
            raise RuntimeError("Call `model.precompute_dataset_rho(...)` first!")
        batch = as_dense_batch(data, self.max_nodes)
        eps = torch.randn(batch.num_graphs, self.z_dim, dtype=torch.float32, device=batch.device)
        return self.forward_with_eps(batch, eps, legacy=True)

    def forward_individual_rho(self, data):
        """The variant kept in the reference's graphvae/models/model_legacy.py:257-363, on this
        model's heads (its decoder `dec` is `dec_edge`; `dec_node` takes no part and receives a
        zero gradient): the matching runs over all max_nodes rows of the padded matrices, the
        ground truth is permuted while the prediction stays in place, and positives are weighted
        by each graph's own edge density.  The only variant whose loss depends on the permutation.
        Needs no `precompute_dataset_rho`."""
        batch = as_dense_batch(data, self.max_nodes)
        eps = torch.randn(batch.num_graphs, self.z_dim, dtype=torch.float32, device=batch.device)
        return self.forward_with_eps(batch, eps, individual_rho=True)

    @torch.no_grad()
    def sample_decode(self, z: torch.Tensor, th_edge: float = 0.5, th_node: float = 0.5) -> dict:
        """Batched decoder pass of the sampler (reference experiments/sample.py:50-73) in one
        library call: logits of `dec_edge` / `dec_node`, thresholded adjacency (dummy rows,
        dummy columns and the diagonal suppressed), node-existence mask and node counts."""
        from ..engine import sample_decode
        if not self._flat.is_cuda:
            raise _hip.GvaeHipError("GraphVAE runs on the GPU only: move the model with .to('cuda')")
        return sample_decode(self.in_dim, self.hid_dim, self.z_dim, self.max_nodes, self._flat,
                             z.to(self._flat.device, torch.float32), th_edge, th_node)

    def last_permutations(self) -> torch.Tensor:
        """perm_full of every graph of the most recent forward, (B, max_nodes) int64."""
        return self._engine.perm.clone()

    def last_mpm_iterations(self) -> torch.Tensor:
        return self._engine.iters_done.clone()
