"""Child process of tests/test_gpu_dist.py: one rank of a two-rank data-parallel run in which
both ranks drive cuda:0 and the collectives go through gloo (what a one-GPU box can rehearse).
usage: python tests/dist_worker.py RANK WORLD PORT OUT_DIR MATCHING MODE STEPS"""
import faulthandler
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "graph-vae_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make(seed_model, matching, steps):
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    torch.manual_seed(seed_model)
    model = GraphVAE(1, 32, 16, 12, matching=matching)
    full = synthetic_batch("enzymes", 8, 12, "deg", seed=11)
    model.precompute_dataset_rho([full])
    eps = torch.randn(steps, 8, 16, generator=torch.Generator().manual_seed(5))
    return model, full, eps


def run_steps(model, full, eps, idx, world, mode, steps, dev):
    from graphvae.train.train_ddp import TrainStep
    model = model.to(dev).train()
    step = TrainStep(model, 1e-2, 1e-4, 5.0, world, overlap_matching=mode != "sequential",
                     pipeline=mode == "pipeline")
    batch = full.index(idx).to(dev)
    for s in range(steps):
        faulthandler.dump_traceback_later(40, exit=True)      # re-armed per step: a stall ends the rank
        step(batch, beta=0.3, lr_scale=min(1.0, s / 2), eps=eps[s][idx].to(dev))
    step.wait()
    torch.cuda.synchronize()
    faulthandler.cancel_dump_traceback_later()
    model._engine.check_matching()
    return model, step


def main():
    rank, world, port, out_dir, matching, mode, steps = sys.argv[1:8]
    rank, world, steps = int(rank), int(world), int(steps)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    faulthandler.dump_traceback_later(60, exit=True)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphvae.train.train_ddp import HostStagedCollectives, shard_indices
        model, full, eps = make(3 + rank, matching, steps)      # ranks start different on purpose
        model = model.to(dev)
        comm = HostStagedCollectives()
        comm.broadcast(model.flat_parameters(), src=0)         # as _ddp_worker does
        comm.broadcast(model.bn_state(), src=0)
        idx = shard_indices(full.num_graphs, world, rank, shuffle=False, epoch=0)
        model, step = run_steps(model, full, eps, idx, world, mode, steps, dev)
        flat = model.flat_parameters().cpu()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "ranks diverged"
        np.save(os.path.join(out_dir, f"params_{rank}.npy"), flat.numpy())
        np.save(os.path.join(out_dir, f"gnorm_{rank}.npy"), step.grad_norm.cpu().numpy())
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
