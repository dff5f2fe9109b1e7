"""world_size-2 test of the data-parallel step on CPU (gloo): graph-batch sharding, buffer
broadcast, one SUM all-reduce of the flat gradient buffer scaled by 1/world, identical
parameters on every rank, and equality with a single process on the union batch.

The HIP engine cannot run here, so TrainStep gets a CPU checker engine built on
oracle/ref_step.py; what is under test is TrainStep's collective pattern and sharding, which
is the same code the GPU ranks execute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_step as R


class OracleEngine:
    """StepEngine look-alike on CPU tensors (checker only)."""

    def __init__(self, model):
        self.model = model
        self.iters_done = torch.zeros(1, dtype=torch.int32)

    def _state(self, params_flat, bn_state):
        m = self.model
        st = {n: v.detach().clone().requires_grad_(True)
              for n, v in zip(R.PARAM_NAMES, m._views_of(params_flat))}
        H = m.hid_dim
        for k, name in enumerate(("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var")):
            st[name] = bn_state[k * H:(k + 1) * H].clone()
        st["bn1.num_batches_tracked"] = torch.tensor(0)
        st["bn2.num_batches_tracked"] = torch.tensor(0)
        return st

    def forward(self, params, bn_state, batch, eps, pos_weight, train):
        st = self._state(params, bn_state)
        st["pos_weight_global"] = pos_weight.reshape(()).clone()
        graphs = R.graphs_from_arrays(batch.x, batch.adj_dense, batch.n_x, batch.num_real_nodes)
        self.res = R.forward_batch(st, graphs, [eps[g:g + 1] for g in range(len(graphs))], train=train,
                                   no_grad_matching=True)
        self.st = st
        H = self.model.hid_dim
        for k, name in enumerate(("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var")):
            bn_state[k * H:(k + 1) * H].copy_(st[name])
        r = self.res
        return torch.stack([r.rec.detach(), r.kl.detach(), r.max_logit, r.logvar_max])

    def backward(self, params, pos_weight, grads, c_rec=1.0, c_kl=1.0, d_up=None):
        (c_rec * self.res.rec + c_kl * self.res.kl).backward()
        for n, view in zip(R.PARAM_NAMES, self.model._views_of(grads)):
            view.copy_(self.st[n].grad)

    def adam_step(self, params, grads, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                  clip_max=0.0, grad_scale=1.0, norm_out=None):
        g = grads * grad_scale
        nrm = g.norm()
        if norm_out is not None:
            norm_out.fill_(float(nrm))
        if clip_max > 0:
            g = g * min(1.0, clip_max / (float(nrm) + 1e-6))
        g = g + weight_decay * params
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
        params.sub_((lr / bc1) * m / (v.sqrt() / bc2 ** 0.5 + eps))


def _make(seed_model=3):
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    torch.manual_seed(seed_model)
    model = GraphVAE(1, 16, 8, 10)
    full = synthetic_batch("enzymes", 8, 10, "deg", seed=11)
    model.precompute_dataset_rho([full])
    eps = torch.randn(3, 8, 8, generator=torch.Generator().manual_seed(5))
    return model, full, eps


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphvae.train.train_ddp import TrainStep, shard_indices
        torch.set_num_threads(2)
        model, full, eps = _make(seed_model=3 + rank)      # ranks start different on purpose
        dist.broadcast(model.flat_parameters(), src=0)     # what _ddp_worker does after building the model
        dist.broadcast(model.bn_state(), src=0)
        step = TrainStep(model, 1e-2, 1e-4, 5.0, world, engine=OracleEngine(model))
        model.train()
        for s in range(3):
            idx = shard_indices(full.num_graphs, world, rank, shuffle=False, epoch=0)
            out4 = step(full.index(idx), beta=0.3, lr_scale=min(1.0, s / 2), eps=eps[s][idx])
        flat = model.flat_parameters().clone()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
        if rank == 0:
            np.save(os.path.join(out_dir, "params.npy"), flat.numpy())
            np.save(os.path.join(out_dir, "gnorm.npy"), step.grad_norm.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process(tmp_path):
    from graphvae.train.train_ddp import TrainStep
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    dp = np.load(tmp_path / "params.npy")
    # single process on the union batch with the same noise
    model, full, eps = _make(seed_model=3)
    step = TrainStep(model, 1e-2, 1e-4, 5.0, 1, engine=OracleEngine(model))
    model.train()
    for s in range(3):
        step(full, beta=0.3, lr_scale=min(1.0, s / 2), eps=eps[s])
    sp = model.flat_parameters().numpy()
    # mean over ranks of per-rank means == mean over the union (equal shard sizes); fp32 sums differ
    # in order only.  The two GCN biases have analytically zero gradient (BatchNorm cancels them), so
    # Adam moves them by +-lr on rounding noise: exclude them (see golden_util.check_params_after_adam).
    from graphvae.engine import PARAM_NAMES
    mask = np.ones_like(sp, dtype=bool)
    for n, v in zip(PARAM_NAMES, model._views_of(torch.arange(sp.size, dtype=torch.float32))):
        if n in ("conv1.bias", "conv2.bias"):
            mask[v.reshape(-1).long().numpy()] = False
    assert np.abs(dp - sp)[mask].max() <= 2e-4 * np.abs(sp).max()
    assert abs(float(np.load(tmp_path / "gnorm.npy")[0]) - float(step.grad_norm)) <= 1e-3 * float(step.grad_norm)
