#!/usr/bin/env python
"""Benchmark of the GraphVAE training step on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c1|c2|c3|c4|c5] [--matching stream|fused]

One process per GPU.  Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this
process is one rank; without it, `--gpus N` with N > 1 makes this process a launcher that starts N
fresh rank processes itself before it touches the GPU (one worker per GPU, as the reference
trainer spawns them, graphvae/train/train_ddp.py:417-421), forwards rank 0's JSON line as its
only stdout and exits with the first non-zero rank status.  A step is one
full optimisation step on one batch of synthetic graphs already resident in HBM: encoder,
heads, decoder, matching (similarity tensor + MPM), assignment, losses, backward, gradient
all-reduce (RCCL), clipping and Adam.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs).  By default c2 keeps the per-GPU batch fixed ("scaling":
"weak") and the others divide their global batch over the ranks ("strong"); `--scaling` selects
either for any workload (weak: every GPU gets the configuration's whole batch, strong: the
configuration's batch is divided over the GPUs):
  c1  ENZYMES-shaped graphs of at most 40 nodes (reference configs/enzymes_deg.yaml:5), batch 32:
      the reference's own CPU-runnable configuration
  c2  dense graphs, max_nodes 32, batch 256 per GPU         (default; matching-dominated)
  c3  ENZYMES-shaped graphs padded to 126, batch 64 over all ranks
  c4  dense graphs, max_nodes 64, batch 1024 over all ranks, fixed MPM iterations (--mpm-iters)
  c5  dense graphs, max_nodes 128, batch 512 over all ranks
The default matching path is "stream": the reference's algorithm with the (R,R,R,R)
similarity tensor materialised in HBM and streamed once per MPM iteration; that is the
kernel the HBM roofline figure refers to.  The S-free fused path gives identical bits and is
timed as well (reported under "fused").  Steps run as the trainer runs them: matching on a
second stream beside backward + Adam, assignment + losses on a third (S-free matching: the whole
matching half as one chain rotating over three streams), up to four steps in flight
(GVAE_BENCH_PIPELINE=0: one step at a time); the timed region ends with a device-wide
synchronisation, so every stream's work of the K steps is inside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "graph-vae_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# RCCL between processes needs dmabuf IPC handles on this driver (hipIpcGetMemHandle fails with the
# legacy mode); set before anything initialises the GPU, inherited by the ranks this script starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
VALU_F32_PEAK_TFLOPS = 157.3  # 256 CUs x 4 SIMDs x 32 fp32 FMA lanes x 2 flop x 2.4 GHz

WORKLOADS = {
    #        kind       N    global batch (None = per-GPU 256)   early_stop
    "c1": ("enzymes_filtered", 40, 32, True),
    "c2": ("dense", 32, None, True),
    "c3": ("enzymes", 126, 64, True),
    "c4": ("dense", 64, 1024, False),
    "c5": ("dense", 128, 512, True),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--matching", choices=["stream", "fused"], default="stream")
    ap.add_argument("--mpm-iters", type=int, default=50)
    ap.add_argument("--per-gpu-batch", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: per-GPU batch fixed as --gpus grows; strong: global batch fixed "
                         "(default: weak for c2, strong for the other workloads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-leg", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def scaling_of(args) -> str:
    return args.scaling or ("weak" if WORKLOADS[args.workload][2] is None else "strong")


def per_gpu_batch(workload: str, scaling: str, world: int) -> int:
    """Graphs per GPU and step: the configuration's batch on every GPU (weak) or divided
    over the GPUs (strong)."""
    config_batch = WORKLOADS[workload][2] or 256
    return config_batch if scaling == "weak" else max(1, config_batch // world)


def launch_ranks(n: int) -> int:
    """Launcher for `--gpus N` outside torchrun: N fresh rank processes of this script, one per
    GPU, started before this process has made any GPU call.  Rank 0's stdout (the JSON line) is
    this process's stdout; the other ranks' stdout goes to stderr.  Returns the exit status."""
    import socket
    import subprocess
    backend = os.environ.get("GVAE_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()          # counts devices without initialising them
    if n_dev == 0:
        print("bench.py needs a GPU (the HIP step has no CPU fallback)", file=sys.stderr)
        return 2
    if backend == "nccl" and n > n_dev:
        print(f"bench.py: --gpus {n} but only {n_dev} visible GPU(s): RCCL needs one GPU per rank "
              "(GVAE_DIST_BACKEND=gloo rehearses more ranks than GPUs)", file=sys.stderr)
        return 2
    with socket.socket() as sk:                # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", str(Path(__file__).resolve())] + sys.argv[1:],
                                      env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            try:
                code = procs[r].wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks",
                      file=sys.stderr)
                for q in pending:              # exactly the processes started above
                    procs[q].terminate()
    return rc


def make_model_and_batches(args, rank, world, device, matching, n_batches=2):
    from graphvae.data import synthetic_batch
    from graphvae.models import GraphVAE
    kind, N, global_batch, early_stop = WORKLOADS[args.workload]
    per_gpu = args.per_gpu_batch or per_gpu_batch(args.workload, scaling_of(args), world)
    torch.manual_seed(42)
    model = GraphVAE(1, 128, 64, N, matching=matching, mpm_iters=args.mpm_iters,
                     mpm_early_stop=early_stop)
    host = [synthetic_batch(kind, per_gpu, N, "deg", seed=1000 * rank + b) for b in range(n_batches)]
    model.precompute_dataset_rho(host)
    model = model.to(device).train()
    return model, [b.to(device) for b in host], host, per_gpu, N, early_stop


def timed_steps(step_fn, batches, steps, warmup, world, device, iters_acc=None, engine=None, flops_acc=None):
    for s in range(warmup):
        step_fn(batches[s % len(batches)], 0.25, 1.0)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        step_fn(batches[s % len(batches)], 0.25, 1.0)
        if iters_acc is not None:
            step_fn.wait()                      # the iteration counts come from the side stream
            iters_acc += engine.iters_done.sum()
            if flops_acc is not None:           # 4 R^3 per executed MPM iteration of a graph
                r3 = batches[s % len(batches)].num_real_nodes.to(torch.float64) ** 3
                flops_acc += (engine.iters_done.to(torch.float64) * 4.0 * r3).sum()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        on = device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(args, host_batches, early_stop):
    """Reference-style CPU path (oracle/ref_step.py: one graph at a time, S materialised,
    autograd on, SciPy assignment, clip + Adam) on a bounded sample of the same workload."""
    from oracle import ref_step as R
    hb = host_batches[0]
    N, B = hb.max_nodes, hb.num_graphs
    st = R.make_trainable(R.init_state(1, 128, 64, N, seed=42))
    opt = torch.optim.Adam([st[n] for n in R.PARAM_NAMES], lr=1e-4, weight_decay=1e-4)
    graphs = R.graphs_from_arrays(hb.x, hb.adj_dense, hb.n_x, hb.num_real_nodes)
    no_grad_matching = N > 48          # the autograd record of 50 R^4 products does not fit
    kw = dict(mpm_iters=args.mpm_iters, early_stop=early_stop, no_grad_matching=no_grad_matching)
    gen = torch.Generator().manual_seed(0)

    def run(n):
        eps = [torch.randn(1, 64, generator=gen) for _ in range(n)]
        t0 = time.perf_counter()
        R.train_step(st, opt, graphs[:n], eps, 1, 10, 1e-4, 0.0, 0.5, 5.0, **kw)
        return time.perf_counter() - t0

    n0 = 1 if N > 48 else 2
    t_probe = run(n0)
    n = int(max(n0, min(B, 0.5 * args.cpu_seconds / max(t_probe / n0, 1e-6))))
    t, reps = 0.0, 0
    while t < args.cpu_seconds and reps < 64:          # bounded: about cpu_seconds of CPU work
        t += run(n)
        reps += 1
    n_total = n * reps
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": n_total / t, "unit": "graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} optimisation steps on {n} of the {B} graphs of one {args.workload} batch "
                      f"(per-graph loop, S materialised{', matching under no_grad' if no_grad_matching else ', autograd through matching'}); "
                      f"host CPU: {cpu}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))       # this process stays off the GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP step has no CPU fallback)")
    # one GPU per rank; GVAE_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 code path
    # on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device)
    backend = os.environ.get("GVAE_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # stdout carries the JSON line and nothing else: until it is printed, file descriptor 1
    # points at stderr, which also catches what native libraries write (RCCL's version banner)
    sys.stdout.flush()
    real_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    # GVAE_BENCH_FORCE_DIST=1: a process group and the step's collectives even with one rank
    force_dist = world == 1 and os.environ.get("GVAE_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29541")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from graphvae.train.train_ddp import require_one_gpu_per_rank
        try:
            require_one_gpu_per_rank(world, backend)       # RCCL: two ranks on one device stall
        except RuntimeError as e:
            raise SystemExit(f"bench.py: {e}")
        if backend == "nccl":                              # "nccl" is RCCL on ROCm
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from graphvae import _hip
    from graphvae.train.train_ddp import TrainStep, require_default_priority

    def note(msg):
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    if os.environ.get("GVAE_HANDOFF_MB"):            # tuning: tail hand-off threshold of the streamed matching
        _hip.set_stream_handoff_bytes(int(float(os.environ["GVAE_HANDOFF_MB"]) * (1 << 20)))
    results = {}
    roof = None
    host_batches = None
    for leg in ([args.matching] + ([] if args.no_fused_leg or args.matching == "fused" else ["fused"])):
        model, batches, host, per_gpu, N, early_stop = make_model_and_batches(args, rank, world, device, leg)
        if leg == "fused" and not _hip.mpm_fused_supported(max(max(b.r_host) for b in batches)):
            continue
        host_batches = host_batches or host
        note(f"{leg}: model and batches ready")
        if dist.is_initialized():
            from graphvae.train.train_ddp import HostStagedCollectives
            HostStagedCollectives().broadcast(model.flat_parameters(), src=0)
            note("parameters broadcast")
        # as the trainer runs it: matching half on a second stream, steps pipelined
        # the step's streams, this one included, must be default-priority streams (event waits
        # between priority classes have stalled the device): fail before the first step, not in it
        require_default_priority([torch.cuda.current_stream(device)], "bench.py")
        step_fn = TrainStep(model, 1e-4, 1e-4, 5.0, world, force_collectives=force_dist,
                            pipeline=os.environ.get("GVAE_BENCH_PIPELINE", "1") == "1")
        eng = model._engine_for(batches[0])
        iters_acc = torch.zeros((), dtype=torch.int64, device=device)
        # W warm-up steps, then EXACTLY K timed steps (no instrumentation) -> `value`
        timed_steps(step_fn, batches, 0, args.warmup, world, device)
        note(f"warm-up done; side stream probe (candidates tried, found): "
             f"{getattr(__import__('graphvae.train.train_ddp', fromlist=['x']).concurrent_stream, 'last_probe', None)}")
        dt = timed_steps(step_fn, batches, args.steps, 0, world, device)
        note(f"{args.steps} timed steps: {1e3 * dt / max(1, args.steps):.3f} ms per step")
        if leg == "stream":
            # roofline pass: the same K steps again with HIP events recorded around every
            # launch of the streaming MPM kernel (on the stream it is launched on)
            _hip.profile_reset()
            _hip.profile_enable(True)
            def stream_counters(reset):
                # (over the workspaces that exist NOW: the instrumented pass may be the first to use
                # the last one of the rotation, whose counters start at zero with its allocation)
                tot_b = tot_n = 0
                cur = eng._cur
                for k in range(len(eng._slots)):
                    eng.use_slot(k)
                    if eng._fwd_batch is not None:
                        b_, n_ = eng.stream_stats(reset=reset)
                        tot_b, tot_n = tot_b + b_, tot_n + n_
                eng.use_slot(cur)
                return tot_b, tot_n

            step_fn.wait()
            torch.cuda.synchronize()
            stream_counters(reset=True)
            timed_steps(step_fn, batches, args.steps, 0, world, device, iters_acc, eng)
            _hip.profile_enable(False)
            ms_iter, n_iter = _hip.profile_read("mpm_stream_iter")
            ms_es, n_es = _hip.profile_read("edge_sim")
            ms_tail, n_tail = _hip.profile_read("mpm_fused")
            _hip.profile_reset()
            # what the streaming kernel executed, counted on the device by the kernel itself:
            # sum over its graph-iterations of 4 R^4 + 12 R^2 bytes
            dev_bytes, dev_gi = stream_counters(reset=False)
            # algorithmic bytes of the streaming MPM kernel: every (graph, iteration) it executes
            # reads S once (4 R^4) plus X, X_new and the node block (12 R^2).  The first iteration
            # of a graph is not among them: the step computes it from row maxima of S without
            # reading the tensor back (gvae_match_stream), so the kernel streams iterations 2..k.
            r = np.array([rr for s in range(args.steps) for rr in batches[s % len(batches)].r_host],
                         dtype=np.float64)
            # Neither are the iterations of the latency-bound tail, which the S-free kernel takes
            # over (gvae_set_stream_handoff_bytes): the kernel counts what it streams.
            # (nor the second one of graphs of at most 64 or at least 89 nodes, which the kernel that produces the
            # tensor computes from the values on their way to memory)
            not_streamed = np.minimum(np.where((r <= 64) | (r >= 89), 2.0, 1.0), float(args.mpm_iters))
            streamed_gi, alg_bytes = float(dev_gi), float(dev_bytes)
            if not early_stop:      # cross-check of the device counters where the count is known
                assert streamed_gi == float(np.sum(args.mpm_iters - not_streamed)), (streamed_gi, len(r))
            achieved = alg_bytes / (ms_iter * 1e-3) / 1e9 if ms_iter > 0 else 0.0
            # HBM bytes per launch from the PMC counters (FETCH_SIZE doubled as gfx950 needs,
            # + WRITE_SIZE), measured with rocprofv3 on this workload's kernel and stored in
            # profiles/pmc_traffic.json as the ratio of counted to algorithmic bytes of the
            # profiled launches; counters cannot be read from inside this process, so that
            # ratio is applied to the bytes the kernel counted itself in this run (a ragged
            # batch has no fixed size per graph-iteration to scale by)
            traffic, traffic_src = None, None
            try:
                pmc = json.load(open(ROOT / "profiles" / "pmc_traffic.json")).get(args.workload)
                if pmc:
                    traffic = pmc["ratio_to_algorithmic"] * alg_bytes / max(1, n_iter)
                    traffic_src = ("profiles/pmc_traffic.json ratio_to_algorithmic (rocprofv3 --pmc FETCH_SIZE / "
                                   "WRITE_SIZE, separate passes) x the bytes this run's kernel counted")
            except OSError:
                pass
            roof = {"kernel": "mpm_stream_kernel", "bound": "hbm", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "launches": int(n_iter), "avg_launch_ms": ms_iter / max(1, n_iter),
                    "alg_bytes_per_launch": alg_bytes / max(1, n_iter),
                    "mean_mpm_iters_per_graph": float(iters_acc.item()) / (args.steps * per_gpu),
                    "streamed_iters_per_graph": streamed_gi / (args.steps * per_gpu),
                    "tail_sfree_ms_per_step": ms_tail / max(1, args.steps),
                    "edge_sim_GBps": (float(np.sum(4.0 * r ** 4)) / (ms_es * 1e-3) / 1e9) if ms_es > 0 else None}
            # step-level matching figure: algorithmic bytes of the kernel that writes the tensor
            # (4 R^4 written + 8 R^2 read per graph) plus those of the streaming kernel, over the
            # time of both
            es_bytes = float(np.sum(4.0 * r ** 4 + 8.0 * r ** 2))
            ms_both = ms_es + ms_iter
            both = (es_bytes + alg_bytes) / (ms_both * 1e-3) / 1e9 if ms_both > 0 else 0.0
            roof["matching_step"] = {"kernels": "edge_sim(+fused second iteration) + mpm_stream_kernel",
                                     "bound": "hbm", "achieved": both, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": both / HBM_PEAK_GBS,
                                     "alg_bytes_per_step": (es_bytes + alg_bytes) / max(1, args.steps),
                                     "ms_per_step": ms_both / max(1, args.steps)}
        fused_roof = None
        if leg == "fused":
            # the S-free kernel is bound by vector fp32 work (4 R^3 flop per graph-iteration, operands
            # resident in LDS), not by HBM: the same K steps again with events around its launches
            _hip.profile_reset()
            _hip.profile_enable(True)
            step_fn.wait()
            torch.cuda.synchronize()
            flops_acc = torch.zeros((), dtype=torch.float64, device=device)
            # (the timed steps ran the assignment as the MPM kernel's epilogue; here the two are
            # separate launches, so that the events bracket the MPM iterations alone)
            step_fn.fuse_assignment = False
            timed_steps(step_fn, batches, args.steps, 0, world, device, iters_acc, eng, flops_acc)
            step_fn.fuse_assignment = True
            _hip.profile_enable(False)
            ms_f, n_f = _hip.profile_read("mpm_fused")
            _hip.profile_reset()
            tf = float(flops_acc.item()) / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
            fused_roof = {"kernel": "mpm_fused_kernel (alone: the timed steps run the assignment as its epilogue)", "bound": "valu", "achieved": tf,
                          "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_F32_PEAK_TFLOPS,
                          "traffic": None, "launches": int(n_f), "avg_launch_ms": ms_f / max(1, n_f),
                          "flop_per_launch": float(flops_acc.item()) / max(1, n_f),
                          "mean_mpm_iters_per_graph": float(iters_acc.item()) / (args.steps * per_gpu)}
        results[leg] = {"value": world * per_gpu * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                        "per_gpu_batch": per_gpu, "roofline": fused_roof}
        del model, batches, step_fn, eng
        torch.cuda.empty_cache()

    main_leg = results[args.matching]
    kind, N, global_batch, early_stop = WORKLOADS[args.workload]
    line = {
        "metric": "training graphs/sec", "value": main_leg["value"], "unit": "graphs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True, "scaling": scaling_of(args),
        "ranks": dist.get_world_size() if dist.is_initialized() else 1,
        "backend": dist.get_backend() if dist.is_initialized() else None,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: synthetic {kind} graphs, max_nodes={N}, "
                               f"batch {main_leg['per_gpu_batch']} per GPU, deg features, hid 128, z 64, "
                               f"MPM iters<={args.mpm_iters} early_stop={early_stop}",
                   "matching": args.matching, "global_batch": main_leg["per_gpu_batch"] * world,
                   "parallelism": f"dp{world}",
                   # the reference's precision everywhere at the boundary; inside, the encoder
                   # forward (GCN layers, BatchNorm, pooling) is evaluated in fp64
                   "precision": "fp32 tensors and matching; encoder forward in fp64"},
    }
    if roof is not None:
        line["roofline"] = roof
    if "fused" in results and args.matching != "fused":
        line["fused"] = {"value": results["fused"]["value"], "ms_per_step": results["fused"]["ms_per_step"],
                         "roofline": results["fused"]["roofline"],
                         "note": "same step with the S-free LDS-resident MPM (bit-identical results)"}
    elif args.matching == "fused":
        line["roofline"] = main_leg["roofline"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # all host threads torch was given, then one thread; each about --cpu-seconds of CPU work
        line["cpu_baseline"] = cpu_baseline(args, host_batches, early_stop)
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            line["cpu_baseline_1thread"] = cpu_baseline(args, host_batches, early_stop)
        finally:
            torch.set_num_threads(threads)
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout_fd, 1)
    os.close(real_stdout_fd)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
