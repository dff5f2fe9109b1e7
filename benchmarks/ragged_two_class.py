"""Experiment: the streamed MPM of an ENZYMES-shaped ragged batch as ONE launch against TWO
concurrent launches on two streams, large graphs (more than `split` nodes) in one, small graphs in
the other, so that the chains of the large non-converging graphs do not queue behind the small
graphs' items.  Matching only (similarity tensors written beforehand).
usage: python benchmarks/ragged_two_class.py [graphs=64] [split=48] [seed=1000]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402
from graphvae.data import synthetic_batch  # noqa: E402
from graphvae.train.train_ddp import concurrent_stream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
split = int(sys.argv[2]) if len(sys.argv) > 2 else 48
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
N = 126
dev = torch.device("cuda:0")
batch = synthetic_batch("enzymes", B, N, "deg", seed=seed).to(dev)
order = sorted(range(B), key=lambda g: -batch.r_host[g])
idx = torch.tensor(order, device=dev)
Rh = [batch.r_host[g] for g in order]
A = batch.adj_dense[idx].contiguous()
gen = torch.Generator().manual_seed(0)
lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
R = torch.tensor(Rh, dtype=torch.int32, device=dev)
NS = _hip.node_sim(A, P, R)
n_big = sum(r > split for r in Rh)
print(f"{B} graphs, sizes {Rh[:6]} ... {Rh[-3:]}, {n_big} above {split} nodes")


def prepare(lo, hi):
    rh = Rh[lo:hi]
    S, S_off = _hip.edge_sim(A[lo:hi], P[lo:hi], NS[lo:hi], R[lo:hi], rh, _hip.S_PADDED)
    ws = torch.empty(_hip.lib().gvae_mpm_stream_ws_bytes(hi - lo, N), dtype=torch.uint8, device=dev)
    return S, S_off, ws, rh


def run(part, lo, hi):
    S, S_off, ws, rh = part
    X = _hip.x0_uniform(rh, N, dev)
    return _hip.mpm_stream(S, S_off, R[lo:hi], max(rh), X, 50, 1e-4, True, ws=ws, layout=_hip.S_PADDED), X


whole = prepare(0, B)
big, small = prepare(0, n_big), prepare(n_big, B)
side = concurrent_stream(dev)
main = torch.cuda.current_stream(dev)


def one():
    return run(whole, 0, B)


def two():
    ev = torch.cuda.Event()
    ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        b = run(big, 0, n_big)
    s = run(small, n_big, B)
    main.wait_stream(side)
    return b, s


for name, fn in (("one launch", one), ("two launches", two)):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{name:14s} {1e3 * min(ts):7.3f} ms (best of 5, includes X0 set-up)")
d1, X1 = one()
(db, Xb), (ds, Xs) = two()
torch.cuda.synchronize()
print("same results:", torch.equal(d1, torch.cat([db, ds])) and torch.equal(X1, torch.cat([Xb, Xs])))
