"""Host enqueue time per pipelined step against the total time per step (200 free-running
steps, one synchronisation at the end): shows whether the Python side keeps ahead of the GPU.
usage: python benchmarks/host_bound.py"""
import sys, time, torch
sys.path.insert(0, "graph-vae_amd")
from graphvae.data import synthetic_batch
from graphvae.models import GraphVAE
from graphvae.train.train_ddp import TrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(42)
for matching in ("fused", "stream"):
    model = GraphVAE(1, 128, 64, 32, matching=matching)
    host = [synthetic_batch("dense", 256, 32, "deg", seed=b) for b in range(2)]
    model.precompute_dataset_rho(host)
    model = model.to(dev).train()
    batches = [b.to(dev) for b in host]
    step = TrainStep(model, 0.0, 0.0, 5.0, pipeline=True)
    for s in range(10):
        step(batches[s % 2], 0.25, 1.0)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for s in range(n):
        step(batches[s % 2], 0.25, 1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{matching}: host enqueue {1e3*(t1-t0)/n:.3f} ms per step, total {1e3*(t2-t0)/n:.3f} ms per step")

# where the host time of a streamed step goes: wall time inside each engine call
import collections
acc = collections.defaultdict(float)
eng = model._engine
for name in ("forward_model", "forward_match_loss", "backward", "adam_step"):
    fn = getattr(eng, name)

    def timed(*a, _fn=fn, _name=name, **k):
        t = time.perf_counter()
        r = _fn(*a, **k)
        acc[_name + (str(a[0]) if _name == "forward_match_loss" and a else "")] += time.perf_counter() - t
        return r
    setattr(eng, name, timed)
n = 100
for s in range(n):
    step(batches[s % 2], 0.25, 1.0)
torch.cuda.synchronize()
print("stream, host ms per step inside:", {k: round(1e3 * v / n, 3) for k, v in acc.items()})
