#!/usr/bin/env python
"""Data-parallel GraphVAE training on MI355X: one process per GPU, graph batches
sharded across ranks, one RCCL all-reduce of the flat gradient buffer per step.

Entry points mirror the reference trainer (graphvae/train/train_ddp.py):
  python -m graphvae.train.train_ddp --dataset {grid,enzymes,synthetic-dense,synthetic-enzymes}
         --feature_type {id,deg} [--epochs E] [--max_nodes M] [--log_dir D] [--config Y]
         [--no_wandb] [--run_name S] [--debug] [--lr LR] [--batch_size B]        (:426-451)
  train(cfg)                      dict / Namespace, YAML `trainer:` section        (:368-421)
  GraphDataset(graphs, max_nodes, feature_type)                                   (:57-112)
  build_graph_list(dataset, max_nodes) -> (graphs, max_nodes)                     (:132-149)
Outputs `log_dir/params.yaml` and `log_dir/graphvae_ddp.pt` (state_dict with the
reference's keys).  Under torchrun (RANK / WORLD_SIZE in the environment) each process
drives the GPU LOCAL_RANK; otherwise one worker per visible GPU is spawned.

Differences from the reference, on purpose: `--epochs` / `--lr` / `--batch_size` are
accepted on the command line (the reference's own job script passes `--epochs`, which its
parser rejects); rho / pos_weight reach every rank (the reference computes them on rank 0
only, after which other ranks cannot run forward); TensorBoard and W&B are optional.
"""
from __future__ import annotations

import argparse
import itertools
import os
import random
from datetime import datetime
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import yaml

from ..data import (DenseGraphBatch, DenseLoader, GraphData, collate, graph_from_adjacency,
                    graph_from_networkx, synthetic_adjacencies)
from .. import _hip
from ..engine import MATCH_STREAM, PHASE_ASSIGN_LOSS, PHASE_MATCH, PHASE_MPM, PHASE_PREP, PHASE_TAIL
from ..models.model import GraphVAE

SEED = 42


def _set_seed() -> None:
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(SEED)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(SEED)


# --------------------------------------------------------------------------- data
class GraphDataset(torch.utils.data.Dataset):
    """List of graphs (networkx graphs, or (R,R) adjacency arrays) -> padded records with
    `x`, `edge_index`, `adj_dense` (unit diagonal on real nodes), `num_real_nodes`."""

    def __init__(self, graphs, max_nodes: int, feature_type: str = "id") -> None:
        if feature_type not in ("id", "deg"):
            raise ValueError(f"Unknown feature_type: {feature_type}")
        self.graphs = graphs
        self.max_nodes = max_nodes
        self.feature_type = feature_type

    def __len__(self) -> int:
        return len(self.graphs)

    def __getitem__(self, idx) -> GraphData:
        g = self.graphs[idx]
        if isinstance(g, np.ndarray):
            return graph_from_adjacency(g, self.max_nodes, self.feature_type)
        return graph_from_networkx(g, self.max_nodes, self.feature_type)

    def dense(self) -> DenseGraphBatch:
        """The whole dataset as one padded batch on the host."""
        return collate([self[i] for i in range(len(self))], self.max_nodes)


def _num_nodes(g) -> int:
    return g.shape[0] if isinstance(g, np.ndarray) else g.number_of_nodes()


def load_tu_dataset(name: str, root: str = "/tmp/TUD"):
    """TU-format reader for a dataset already on disk (`root/name/raw/name_A.txt`,
    `name_graph_indicator.txt`); nothing is downloaded."""
    raw = Path(root) / name / "raw"
    a_file, gi_file = raw / f"{name}_A.txt", raw / f"{name}_graph_indicator.txt"
    if not a_file.exists() or not gi_file.exists():
        raise FileNotFoundError(
            f"{name}: {a_file} / {gi_file} not found; place the TU dataset files there "
            "(no network download is attempted) or use --dataset synthetic-enzymes")
    gi = np.loadtxt(gi_file, dtype=np.int64)
    edges = np.loadtxt(a_file, delimiter=",", dtype=np.int64)
    n_graphs = int(gi.max())
    first = np.searchsorted(gi, np.arange(1, n_graphs + 1))
    sizes = np.bincount(gi, minlength=n_graphs + 1)[1:]
    adjs = [np.zeros((s, s), np.float32) for s in sizes]
    for u, v in edges:
        g = gi[u - 1] - 1
        adjs[g][u - 1 - first[g], v - 1 - first[g]] = 1.0
    return [np.maximum(a, a.T) for a in adjs]


def build_graph_list(dataset: str, max_nodes: int | None, n_synthetic: int = 600):
    if dataset == "grid":
        import networkx as nx
        graphs = [nx.grid_2d_graph(i, j) for i, j in itertools.product(range(2, 4), repeat=2)]
    elif dataset == "enzymes":
        graphs = load_tu_dataset("ENZYMES")
    elif dataset in ("synthetic-dense", "synthetic-enzymes"):
        kind = dataset.split("-")[1]
        graphs = synthetic_adjacencies(kind, n_synthetic, max_nodes or (32 if kind == "dense" else 126),
                                       seed=0)
    else:
        raise ValueError("dataset must be 'grid', 'enzymes', 'synthetic-dense' or 'synthetic-enzymes'")
    if max_nodes is None:
        max_nodes = max(_num_nodes(g) for g in graphs)
    graphs = [g for g in graphs if _num_nodes(g) <= max_nodes]
    return graphs, max_nodes


def shard_indices(n: int, world: int, rank: int, shuffle: bool, epoch: int, seed: int = 0):
    """Sample order of one rank: same rule as torch's DistributedSampler (pad by
    repetition to a multiple of world, then take rank::world)."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = (n + world - 1) // world * world
    if total > n:
        idx += (idx * ((total - n + n - 1) // n))[: total - n]
    return idx[rank:total:world]


# --------------------------------------------------------------------------- collectives
class HostStagedCollectives:
    """The step's collectives on flat device tensors, for any torch.distributed backend.

    RCCL ("nccl") collectives are stream-ordered device operations and take the tensor as it is.
    Every other backend (gloo) reduces on the host.  c10d's own handling of a device tensor there
    (a high-priority pool stream, pinned temporaries and event waits inside a worker thread)
    stalled within a few steps on MI355X whenever kernels of the step were still in flight when
    the collective was issued: the device work queued ahead of the collective never completed on
    one rank (benchmarks/rehearse_dist.py shows it).  So the ordering is made explicit here
    instead: copy to a pinned buffer on the current stream, synchronise THAT stream, run the
    collective on the host copy, copy back on the same stream."""

    def __init__(self, group=None) -> None:
        self.group = group
        self._pinned = {}

    def _direct(self, t: torch.Tensor) -> bool:
        return not t.is_cuda or dist.get_backend(self.group) == "nccl"

    def _staged(self, t: torch.Tensor, fn) -> None:
        """`t` must be contiguous (the result is written back through a flat view of it) and all
        calls must come from one stream: the pinned buffer of a (size, dtype) is reused."""
        if not t.is_contiguous():
            raise ValueError("HostStagedCollectives: the tensor must be contiguous")
        key = (t.numel(), t.dtype)
        h = self._pinned.get(key)
        if h is None:
            h = self._pinned[key] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
        h.copy_(t.view(-1), non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        fn(h)
        t.view(-1).copy_(h, non_blocking=True)

    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM) -> None:
        if self._direct(t):
            dist.all_reduce(t, op=op, group=self.group)
        else:
            self._staged(t, lambda h: dist.all_reduce(h, op=op, group=self.group))

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        if self._direct(t):
            dist.broadcast(t, src=src, group=self.group)
        else:
            self._staged(t, lambda h: dist.broadcast(h, src=src, group=self.group))


def require_one_gpu_per_rank(world: int, backend: str = "nccl") -> None:
    """RCCL needs a device of its own for every rank of a node: two ranks on one device do not
    fail, they stall inside the first collective.  Raised before any collective is issued."""
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if backend == "nccl" and local_world > n_dev:
        raise RuntimeError(
            f"{local_world} ranks on this node but only {n_dev} visible GPU(s): RCCL needs one GPU "
            "per rank (launch with --nproc-per-node <= GPUs; to rehearse more ranks than GPUs use "
            "the gloo backend)")


# --------------------------------------------------------------------------- the step
def concurrent_stream(device, others=(), tries: int = 4) -> "torch.cuda.Stream":
    """A stream whose kernels run beside the current stream's (and beside those of `others`).
    HIP multiplexes a process's streams over a few hardware queues (four by default) and
    packets of one queue execute in order, so two streams that share a queue run one after the
    other however independent their work is; which pool stream shares a queue with which
    depends on how many streams other libraries (RCCL, c10d) took before.  Up to `tries`
    consecutive pool streams are probed: about 2 ms of memory-bound kernels go to a reference
    stream, a one-element kernel to the candidate; the candidate passes if its kernel ends
    while the others still run.  When no candidate passes (or tries == 0) the last pool stream
    is returned all the same: the step is then correct but its halves run one after the other.
    Only default-priority streams are handed out.  (A high-priority stream also gets a queue of
    its own, but with this step's kernels in flight, event waits between a high-priority and a
    default-priority stream stalled the device within a few steps, although the event graph
    has no cycle; TrainStep asserts the priority of the streams it is given.)"""
    refs = [torch.cuda.current_stream(device)] + list(others)
    ballast = torch.empty(64 << 20, dtype=torch.float32, device=device)    # 256 MiB
    flag = torch.zeros(1, device=device)
    held = []                       # keeps rejected candidates alive so the pool moves on
    chosen = None
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device)
        held.append(cand)
        ok = True
        for ref in refs:
            torch.cuda.synchronize(device)
            done_ref, done_cand = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(ref):
                for _k in range(20):
                    ballast.add_(1.0)
                done_ref.record(ref)
            with torch.cuda.stream(cand):
                flag.add_(1.0)
                done_cand.record(cand)
            done_cand.synchronize()
            if done_ref.query():    # the candidate only ran once the ballast was through
                ok = False
                break
        if ok:
            chosen = cand
            break
    torch.cuda.synchronize(device)
    concurrent_stream.last_probe = (len(held), chosen is not None)     # for logs
    if chosen is None:
        chosen = held[-1] if held else torch.cuda.Stream(device=device)
    return chosen


def require_default_priority(streams, what: str) -> None:
    """Raises if any of the streams (None entries are skipped) is not a default-priority stream.
    The step exchanges event waits between its streams while kernels of several of them are in
    flight; twice on MI355X such waits between a high-priority and a default-priority stream
    ended with the device no longer retiring the default-priority stream's work (a side stream
    created with priority -1, and c10d's gloo path, which stages device tensors on a
    high-priority pool stream), although the event graph had no cycle.  Every stream of a step,
    the caller's current one included, therefore has to be of the default priority class."""
    for st in streams:
        if st is not None and getattr(st, "priority", 0) != 0:
            raise _hip.GvaeHipError(
                f"{what}: stream {st} has priority {st.priority}; the step's streams (the current one "
                "included) must all have default priority -- event waits between priority classes "
                "have stalled the device")


class TrainStep:
    """forward + backward + gradient all-reduce + clip + Adam, all on flat buffers.
    loss = rec + beta * kl (reference train_ddp.py:282-292,316)."""

    def __init__(self, model: GraphVAE, lr: float, weight_decay: float, clip_max: float,
                 world: int = 1, group=None, sync_bn_buffers: bool = False, engine=None,
                 force_collectives: bool = False, overlap_matching: bool = True,
                 pipeline: bool = False, probe_tries: int = 4, slots: int | None = None) -> None:
        """`engine`: object with StepEngine's forward / backward / adam_step; defaults to the
        model's HIP engine (tests inject a CPU checker to exercise the collective pattern).
        `sync_bn_buffers`: broadcast rank 0's BatchNorm running statistics before every forward,
        as DDP's broadcast_buffers does.  Training forwards never read them (batch statistics
        are used) and rank 0's copy depends on rank 0's batches only, so by default the
        broadcast is left to `sync_buffers()`, which the trainer calls before validation and
        before saving: same statistics wherever they are read, one collective fewer per step.
        `force_collectives`: issue the step's collectives even in a group of one rank (a
        single-GPU rehearsal of the stream ordering between RCCL and the HIP kernels).
        `overlap_matching`: run matching + assignment + losses on a second HIP stream while the
        backward, the all-reduce and Adam run on the current one.  The gradients do not depend
        on the permutation (reference model.py:325-334 gathers prediction and ground truth
        through the same perm_full, so the edge loss visits each pair of real nodes exactly
        once), hence both halves only need the decoder's logits; they are joined before the
        step returns.  Results are bit-identical to the sequential order.
        `pipeline` (needs overlap_matching): the step returns WITHOUT that join, so the matching
        half of step N also runs beside the model half, backward and Adam of the following steps
        (four workspaces in rotation: `slots`, or the environment's GVAE_PIPELINE_SLOTS; a workspace
        is reused only after its matching has finished.  With the S-free matching the chain MPM ->
        assignment -> losses of a step is latency-bound and outlives two further steps, so the
        chains of consecutive steps rotate over three streams; with fewer workspaces than chains
        in flight plus one, the forward of a step waits for the assignment of an earlier one:
        0.30 ms per step with two workspaces against 0.25 with three at 256 graphs of 32 nodes
        before the rotation, 0.23 ms with four and the rotation).
        The returned loss tensor, `model.last_permutations()` and `loss_sum` are then valid on
        the current stream only after `wait()`; parameters, gradients and optimiser state never
        depend on the matching and are always in stream order.
        `probe_tries`: pool streams `concurrent_stream` may probe for the side streams."""
        self.model = model
        self.probe_tries = int(probe_tries)
        self._engine = engine
        self.lr, self.weight_decay, self.clip_max = lr, weight_decay, clip_max
        self.world, self.group = world, group
        self.sync_bn_buffers = sync_bn_buffers
        self.collectives = world > 1 or force_collectives
        self.comm = HostStagedCollectives(group)
        flat = model.flat_parameters()
        self.overlap_matching = bool(overlap_matching) and flat.is_cuda
        self.pipeline = bool(pipeline) and self.overlap_matching
        self._side = None       # second stream + events, created on first use
        self._tail = None       # pipelined: third stream for assignment + losses
        self._fourth = None     # pipelined: the S-free tail of a streamed matching / every other assignment
        self._rot = 0
        self.fuse_assignment = True   # S-free pipelined steps: the assignment as the MPM kernel's epilogue
        self._ev_acc_last = self._acc_stream_last = None
        self._ev_model = self._ev_match = self._ev_mpm = self._ev_tail = self._ev_tail_last = None
        self._slot = 0
        self.n_slots = max(2, int(slots if slots is not None else os.environ.get("GVAE_PIPELINE_SLOTS", "4")))
        # sum over the steps since the last pop_loss_sum() of rec + beta * kl, accumulated where
        # the losses are produced (on the side stream when the matching is overlapped)
        self.loss_sum = torch.zeros(1, dtype=torch.float64, device=flat.device)
        # the [rec, kl, max_logit, logvar_max] copies a step returns: a ring allocated once, so that
        # no block of the caching allocator changes hands between streams on the way (a copy is
        # written on a side stream; a block allocated per step on the main stream could still be
        # in use by earlier main-stream work when that write lands)
        self._out_ring = torch.zeros(32, 4, dtype=torch.float32, device=flat.device)
        self._out_next = 0
        self.grads = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.t = 0

    def wait(self) -> None:
        """Orders the current stream behind every matching half still in flight (a stream
        wait, not a host synchronisation).  Needed before reading losses, permutations or
        iteration counts of pipelined steps, and before a forward outside this object."""
        if self._ev_match is not None:
            main = torch.cuda.current_stream(self.grads.device)
            for ev in self._ev_match:
                if ev is not None:
                    main.wait_event(ev)

    def pop_loss_sum(self) -> torch.Tensor:
        """Sum of rec + beta * kl over the steps since the last call (float64, device), then
        reset; waits for the matching halves in flight."""
        self.wait()
        out = self.loss_sum.clone()
        if self._side is not None:
            acc_stream = self._tail if self._tail is not None else self._side
            acc_stream.wait_stream(torch.cuda.current_stream(self.grads.device))
            with torch.cuda.stream(acc_stream):
                self.loss_sum.zero_()
            if self._ev_acc_last is not None:         # the next update may come from another stream
                self._ev_acc_last = torch.cuda.Event()
                self._ev_acc_last.record(acc_stream)
                self._acc_stream_last = acc_stream
        else:
            self.loss_sum.zero_()
        return out

    def sync_buffers(self) -> None:
        """Every rank takes rank 0's BatchNorm running statistics."""
        if self.collectives:
            self.comm.broadcast(self.model.bn_state(), src=0)

    def __call__(self, batch: DenseGraphBatch, beta: float, lr_scale: float,
                 eps: torch.Tensor | None = None) -> torch.Tensor:
        """One optimisation step; returns a copy of [rec, kl, max_logit, logvar_max] (device; with
        the matching on a side stream it is a row of a ring of 32 buffers, overwritten 32 steps
        later).  `eps` (B, z_dim) overrides the reparameterisation noise (tests)."""
        model = self.model
        if self.collectives and self.sync_bn_buffers:
            # DDP's broadcast_buffers: every forward starts from rank 0's running statistics
            self.comm.broadcast(model.bn_state(), src=0)
        eng = self._engine if self._engine is not None else model._engine_for(batch)
        if eps is None:
            eps = torch.randn(batch.num_graphs, model.z_dim, dtype=torch.float32, device=batch.device)
        overlap = self.overlap_matching and hasattr(eng, "forward_match_loss")
        if overlap:
            main = torch.cuda.current_stream(self.grads.device)
            # event waits between priority classes stalled the device (see concurrent_stream)
            require_default_priority([main], "TrainStep")
            if self._side is None:
                self._side = concurrent_stream(self.grads.device, tries=self.probe_tries)
                self._ev_model = [torch.cuda.Event() for _ in range(self.n_slots)]
                self._ev_mpm = [torch.cuda.Event() for _ in range(self.n_slots)]
                self._ev_tail = [torch.cuda.Event() for _ in range(self.n_slots)]
                self._ev_tail_last = None
                self._ev_match = [None] * self.n_slots
                if self.pipeline:
                    # assignment + losses of step N (one wavefront per graph, latency-bound) run
                    # beside the similarity tensor and the MPM of step N+1
                    self._tail = concurrent_stream(self.grads.device, others=[self._side],
                                                   tries=self.probe_tries)
                    self._fourth = concurrent_stream(self.grads.device, others=[self._side, self._tail],
                                                     tries=self.probe_tries)
                require_default_priority([self._side, self._tail, self._fourth], "TrainStep")
            k = self._slot
            if self.pipeline:
                eng.use_slot(k)
                if self._ev_match[k] is not None:
                    # the matching that last used this workspace (two steps ago) is finished
                    main.wait_event(self._ev_match[k])
            eng.forward_model(model.flat_parameters(), model.bn_state(), batch, eps,
                              model._pos_weight_dev(), True)
            self._ev_model[k].record(main)

            def enqueue_matching():
                # (streams are passed to the engine; a stream context costs the host more than
                # the launches it would wrap)
                side, tail = self._side, self._tail
                if tail is None:
                    side.wait_event(self._ev_model[k])
                    res, last = eng.forward_match_loss(stream=side), side
                elif getattr(eng, "last_mode", None) != MATCH_STREAM:
                    # S-free matching: MPM, assignment (the epilogue of a graph's work-group, as
                    # soon as that graph has converged) and losses of a step are one chain, and the
                    # chains of consecutive steps rotate over the three side streams.  Both kernels
                    # are latency-bound (a work-group, then a wavefront, per graph: 0.17 + 0.15 ms
                    # at 32 nodes, 0.55 + 1.1 ms on ENZYMES-shaped batches, whatever the batch
                    # size), so three of them side by side cost a step a third of the chain.
                    last = (side, tail, self._fourth)[self._rot]
                    self._rot = (self._rot + 1) % 3
                    last.wait_event(self._ev_model[k])
                    if self.fuse_assignment:
                        res = eng.forward_match_loss(PHASE_MATCH | PHASE_ASSIGN_LOSS, stream=last)
                    else:       # (the MPM kernel and the assignment as two launches: to time them apart)
                        eng.forward_match_loss(PHASE_MATCH, stream=last)
                        res = eng.forward_match_loss(PHASE_ASSIGN_LOSS, stream=last)
                else:
                    # Streamed matching: the similarity tensor of this step is written while the
                    # S-free tail of the previous step (a work-group per surviving graph, on a
                    # stream of its own) still runs; the persistent launch waits for that tail, which
                    # would find no room beside it
                    side.wait_event(self._ev_model[k])
                    eng.forward_match_loss(PHASE_PREP, stream=side)
                    if self._ev_tail_last is not None:
                        side.wait_event(self._ev_tail_last)
                    eng.forward_match_loss(PHASE_MPM, stream=side)
                    self._ev_mpm[k].record(side)
                    # (a stream of its own: on the third one it would queue behind the previous
                    # step's assignment, which runs beside this step's persistent launch and as long)
                    fourth = self._fourth
                    fourth.wait_event(self._ev_mpm[k])
                    eng.forward_match_loss(PHASE_TAIL, stream=fourth)
                    self._ev_tail[k].record(fourth)
                    self._ev_tail_last = self._ev_tail[k]
                    tail.wait_event(self._ev_tail[k])
                    res, last = eng.forward_match_loss(PHASE_ASSIGN_LOSS, stream=tail), tail
                out = self._out_ring[self._out_next]  # written on `last`; reused 32 steps later
                self._out_next = (self._out_next + 1) % self._out_ring.shape[0]
                if self._ev_acc_last is not None and self._acc_stream_last is not last:
                    last.wait_event(self._ev_acc_last)       # the loss sum is updated in step order
                if hasattr(eng, "accumulate_loss"):
                    eng.accumulate_loss(res, beta, self.loss_sum, copy=out, stream=last)
                else:
                    with torch.cuda.stream(last):
                        out.copy_(res)
                        self.loss_sum += (out[0] + beta * out[1]).double()
                if self._ev_match[k] is None:
                    self._ev_match[k] = torch.cuda.Event()
                self._ev_match[k].record(last)
                self._ev_acc_last, self._acc_stream_last = self._ev_match[k], last
                return out

            # pipelined: the main chain is enqueued first (the runtime makes the host wait once a
            # stream holds more than a few dozen unfinished commands, which the side streams do
            # while the previous MPM runs; that wait must not hold back the backward)
            if not self.pipeline:
                out4 = enqueue_matching()
        else:
            out4 = eng.forward(model.flat_parameters(), model.bn_state(), batch, eps,
                               model._pos_weight_dev(), True).clone()
            self.loss_sum += (out4[0] + beta * out4[1]).double()
        eng.backward(model.flat_parameters(), model._pos_weight_dev(), self.grads, 1.0, beta)
        if self.collectives:
            self.comm.all_reduce(self.grads)
        self.t += 1
        eng.adam_step(model.flat_parameters(), self.grads, self.m, self.v, self.t,
                      self.lr * lr_scale, weight_decay=self.weight_decay, clip_max=self.clip_max,
                      grad_scale=1.0 / self.world, norm_out=self.grad_norm)
        if overlap and self.pipeline:
            out4 = enqueue_matching()
            self._slot = (self._slot + 1) % self.n_slots
        if overlap and not self.pipeline:
            # join: what follows on this stream (the caller reading the losses, the next forward
            # overwriting the workspace) is ordered behind the matching half
            main.wait_event(self._ev_match[k])
        torch._foreach_add_([model.bn1.num_batches_tracked, model.bn2.num_batches_tracked],
                            batch.num_graphs)
        return out4


class _NoDbg:
    step = 0

    def log(self, *_, **__):
        pass

    def next(self):
        self.step += 1


class DebugState(_NoDbg):
    """Per-step scalars to TensorBoard / W&B when those packages are present."""

    def __init__(self, writer, wandb_run) -> None:
        self.tb, self.wb = writer, wandb_run

    def log(self, tag, value):
        if self.tb is not None:
            self.tb.add_scalar(tag, value, self.step)
        if self.wb is not None:
            self.wb.log({tag: value, "step": self.step})


def _optional_loggers(args, rank: int):
    tb = wb = None
    if rank != 0:
        return tb, wb
    try:
        from torch.utils.tensorboard import SummaryWriter
        tb = SummaryWriter(Path(args.log_dir, "tb"))
    except Exception:
        tb = None
    if getattr(args, "use_wandb", False) and os.getenv("WANDB_DISABLED", "0") not in ("true", "1"):
        try:
            import wandb
            wb = wandb.init(project="graphvae", name=args.run_name or f"{args.dataset}-ddp",
                            dir=args.log_dir, config=vars(args))
        except Exception:
            wb = None
    return tb, wb


# --------------------------------------------------------------------------- worker
def _ddp_worker(rank: int, world: int, args) -> None:
    _set_seed()
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    use_gpu = torch.cuda.is_available()
    if not use_gpu:
        raise RuntimeError("graphvae.train.train_ddp needs a GPU (the HIP step has no CPU fallback)")
    if dist.is_initialized() and dist.get_backend() != "nccl":
        local_rank %= torch.cuda.device_count()      # gloo rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    spawned = False
    if world > 1 and not dist.is_initialized():
        require_one_gpu_per_rank(world)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # "nccl" is RCCL on ROCm
        spawned = True
    comm = HostStagedCollectives()

    # graphs: every rank builds the same list; rank 0's shuffle order is broadcast
    graphs, max_nodes = build_graph_list(args.dataset, args.max_nodes)
    order = list(range(len(graphs)))
    if rank == 0:
        random.shuffle(order)
    if world > 1:
        box = [order]
        dist.broadcast_object_list(box, src=0)
        order = box[0]
    graphs = [graphs[i] for i in order]
    split = int(0.9 * len(graphs))
    g_train, g_val = graphs[:split], graphs[split:]

    ds_tr = GraphDataset(g_train, max_nodes=max_nodes, feature_type=args.feature_type)
    ds_va = GraphDataset(g_val, max_nodes=max_nodes, feature_type=args.feature_type)
    dense_tr = ds_tr.dense()
    n_tr, n_va = len(ds_tr), len(ds_va)
    dl_tr = DenseLoader(dense_tr, args.batch_size,
                        lambda ep: shard_indices(n_tr, world, rank, True, ep), device)
    dl_va = DenseLoader(ds_va.dense(), args.batch_size,
                        lambda ep: shard_indices(n_va, world, rank, False, ep), device) if n_va else []

    model = GraphVAE(in_dim=args.in_dim, hid_dim=args.hid_dim, z_dim=args.z_dim, max_nodes=max_nodes,
                     pool=args.pool, matching=getattr(args, "matching", "auto")).to(device)
    # rho over the full training set, identical on every rank (same data, same arithmetic)
    model.precompute_dataset_rho([dense_tr])
    if world > 1:
        comm.broadcast(model.flat_parameters(), src=0)
        comm.broadcast(model.bn_state(), src=0)
        dist.barrier()

    step_fn = TrainStep(model, args.lr, args.weight_decay, args.clip_max, world,
                        pipeline=not args.debug)
    tb, wb = _optional_loggers(args, rank)
    dbg = DebugState(tb, wb) if (args.debug and rank == 0) else _NoDbg()

    steps_per_epoch = len(dl_tr)
    anneal_steps = max(1, args.warm_epochs * steps_per_epoch)
    global_step = 0
    acc = torch.zeros(1, dtype=torch.float64, device=device)

    for ep in range(args.epochs):
        dl_tr.set_epoch(ep)
        model.train()
        acc.zero_()
        for batch in dl_tr:
            frac = min(1.0, global_step / anneal_steps)
            beta = args.beta0 + (args.beta_final - args.beta0) * frac
            out4 = step_fn(batch, beta, frac)    # losses summed on the device: no per-step sync
            if args.debug and rank == 0:
                o = out4.tolist()
                dbg.log("train/rec", o[0])
                dbg.log("train/kl", o[1])
                dbg.log("train/loss", o[0] + beta * o[1])
                dbg.log("debug/max_logit", o[2])
                dbg.log("debug/logvar_max", o[3])
                if dbg.step % 10 == 0:
                    dbg.log("debug/param_norm", model.flat_parameters().norm().item())
                    dbg.log("debug/grad_norm", step_fn.grad_norm.item())
            dbg.next()
            global_step += 1
        tot_tr = step_fn.pop_loss_sum()                     # joins the matching halves in flight
        if getattr(model, "_engine", None) is not None:
            model._engine.check_matching()                  # once per epoch: faults of the matching kernels
        if world > 1:
            comm.all_reduce(tot_tr)
        mean_tr = tot_tr.item() / max(1, len(dl_tr)) / world

        step_fn.sync_buffers()          # validation reads the running statistics: rank 0's, everywhere
        model.eval()
        acc.zero_()
        with torch.no_grad():
            for batch in dl_va:
                logs = model(batch)
                acc += (logs["rec"] + args.beta_final * logs["kl"]).double()
        tot_va = acc.clone()
        if world > 1:
            comm.all_reduce(tot_va)
        mean_va = tot_va.item() / max(1, len(dl_va)) / world

        if rank == 0:
            print(f"Ep {ep:03d} | train {mean_tr:.4f} | val {mean_va:.4f}", flush=True)
            if tb is not None:
                tb.add_scalars("loss", {"train": mean_tr, "val": mean_va}, ep)
            if wb is not None:
                wb.log({"loss/train": mean_tr, "loss/val": mean_va, "epoch": ep})

    if rank == 0:
        ckpt = Path(args.log_dir, "graphvae_ddp.pt")
        torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, ckpt)
        print("saved", ckpt)
        if tb is not None:
            tb.close()
        if wb is not None:
            wb.finish()
    if spawned:
        dist.destroy_process_group()


# --------------------------------------------------------------------------- train(cfg)
_DEFAULTS = dict(
    dataset="grid", feature_type="deg", lr=1e-4, epochs=200, max_nodes=None, log_dir="runs",
    use_wandb=True, run_name=None, batch_size=64, weight_decay=1e-4, beta0=0.0, beta_final=0.5,
    warm_epochs=40, clip_max=5.0, hid_dim=128, z_dim=64, pool="sum", debug=False, matching="auto",
)


def train(cfg) -> None:
    # RCCL between processes needs dmabuf IPC handles on this driver (hipIpcGetMemHandle fails in the
    # legacy mode): set before anything initialises the GPU here, inherited by spawned workers
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if isinstance(cfg, dict):
        cfg = SimpleNamespace(**cfg.get("trainer", cfg))
    for k, v in _DEFAULTS.items():
        if getattr(cfg, k, None) is None and k not in ("max_nodes", "run_name"):
            setattr(cfg, k, v)
        elif not hasattr(cfg, k):
            setattr(cfg, k, v)
    for key in ("lr", "weight_decay", "beta0", "beta_final", "clip_max"):
        val = getattr(cfg, key)
        if isinstance(val, str):
            try:
                setattr(cfg, key, float(val))
            except ValueError:
                raise ValueError(f"{key} should be numeric, got {val!r}")

    if cfg.log_dir in ("runs", "./runs"):
        cfg.log_dir = str(Path("runs") / datetime.now().strftime("%Y%m%d_%H%M%S"))
    Path(cfg.log_dir).mkdir(parents=True, exist_ok=True)

    graphs, max_nodes_auto = build_graph_list(cfg.dataset, cfg.max_nodes)
    probe = GraphDataset([graphs[0]], feature_type=cfg.feature_type,
                         max_nodes=cfg.max_nodes or max_nodes_auto)
    cfg.in_dim = int(probe[0].x.size(-1))

    if os.environ.get("RANK", "0") == "0":
        with open(Path(cfg.log_dir) / "params.yaml", "w") as f:
            yaml.safe_dump(vars(cfg), f, sort_keys=False)
            print("saved params to", f.name)

    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world > 1 and not dist.is_initialized():
            backend = os.environ.get("GVAE_DIST_BACKEND", "nccl")    # gloo: rehearsal on fewer GPUs
            require_one_gpu_per_rank(world, backend)
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
                                  % torch.cuda.device_count())
            dist.init_process_group(backend)
            try:
                _ddp_worker(int(os.environ["RANK"]), world, cfg)
            finally:
                dist.destroy_process_group()
        else:
            _ddp_worker(int(os.environ["RANK"]), world, cfg)
    else:
        world = torch.cuda.device_count()
        if world <= 1:
            _ddp_worker(0, 1, cfg)
        else:
            torch.multiprocessing.spawn(_ddp_worker, nprocs=world, args=(world, cfg))


def main(argv=None) -> None:
    pa = argparse.ArgumentParser(description="GraphVAE data-parallel training (MI355X HIP step)")
    pa.add_argument("--dataset", choices=["grid", "enzymes", "synthetic-dense", "synthetic-enzymes"],
                    default=None)
    pa.add_argument("--feature_type", choices=["id", "deg"], default=None)
    pa.add_argument("--epochs", type=int, default=None)
    pa.add_argument("--lr", type=float, default=None)
    pa.add_argument("--batch_size", type=int, default=None)
    pa.add_argument("--max_nodes", type=int, default=None)
    pa.add_argument("--matching", choices=["auto", "stream", "fused"], default=None,
                    help="MPM kernel: stream the materialised similarity tensor, or the fused S-free form")
    pa.add_argument("--log_dir", default="runs")
    pa.add_argument("--config", help="YAML config with a 'trainer:' section")
    pa.add_argument("--no_wandb", action="store_true", help="disable Weights & Biases logging")
    pa.add_argument("--run_name", help="custom W&B run name")
    pa.add_argument("--debug", action="store_true", help="per-step scalars to TB / W&B")
    args = pa.parse_args(argv)
    args.use_wandb = not args.no_wandb
    ycfg = {}
    if args.config:
        with open(args.config) as f:
            ycfg = yaml.safe_load(f).get("trainer", {})
    cli = {k: v for k, v in vars(args).items() if k not in ("config", "no_wandb") and v is not None}
    if not args.debug:
        cli.pop("debug", None)          # a store_true default must not override the YAML value
    if not args.no_wandb:
        cli.pop("use_wandb", None)
    train(SimpleNamespace(**{**ycfg, **cli}))


if __name__ == "__main__":
    main()
