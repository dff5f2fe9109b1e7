"""Timeline of the streamed MPM from a trace build of the library (-DGVAE_STREAM_TRACE):
how many work-groups stream at a time, and where a graph's iteration spends its time once
the device is no longer full.
usage: GVAE_HIP_LIB=<trace build> python benchmarks/stream_trace.py [N] [B] [iters] [early_stop] [enzymes]"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
es = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
ragged = len(sys.argv) > 5 and sys.argv[5] == "enzymes"
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
if ragged:          # ENZYMES-shaped sizes and sparsity, padded to N
    from graphvae.data import synthetic_adjacencies
    adjs = synthetic_adjacencies("enzymes", B, N, seed=1000)
    Rh = [a.shape[0] for a in adjs]
    A = torch.zeros(B, N, N)
    for k, a in enumerate(adjs):
        r = a.shape[0]
        A[k, :r, :r] = torch.from_numpy(a) + torch.eye(r)
    A = A.to(dev)
else:
    up = torch.triu((torch.rand(B, N, N, generator=gen) < 0.5).float(), 1)
    A = (up + up.transpose(1, 2) + torch.eye(N)).to(dev)
    Rh = [N] * B
lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
R = torch.tensor(Rh, dtype=torch.int32, device=dev)
NS = _hip.node_sim(A, P, R)
S, S_off = _hip.edge_sim(A, P, NS, R, Rh, 1)
ws = torch.empty(_hip.lib().gvae_mpm_stream_ws_bytes(B, N), dtype=torch.uint8, device=dev)
X0 = _hip.x0_uniform(Rh, N, dev)
cap = 600_000
buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
L = _hip.lib()
L.gvae_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
for rep in range(2):                       # the second run is the one analysed
    torch.cuda.synchronize()
    assert L.gvae_debug_trace(buf.data_ptr(), cap) == 0
    X = X0.clone()
    done = _hip.mpm_stream(S, S_off, R, N, X, iters, 1e-4, es, ws=ws, layout=1)
    torch.cuda.synchronize()
n = L.gvae_debug_trace_count()
print(f"{n} item records, {int(done.sum())} graph-iterations; sizes max {max(Rh)} mean {sum(Rh)/len(Rh):.1f}; "
      f"algorithmic bytes {sum(int(d) * (4 * r ** 4 + 12 * r * r) for d, r in zip(done.tolist(), Rh)) / 1e9:.2f} GB")
t = buf[: 8 * n].cpu().numpy().reshape(n, 8).astype(np.int64)
item = t[:, 0] & 0xFFFFFFFF
g = item >> 16
rows = t[:, 7] & 0xFFFFFFFF
units = t[:, 7] >> 32
tick = 0.01                                 # us per 100 MHz tick
t0 = t[:, 1].min()
T = (t[:, 1:7] - t0) * tick
T[t[:, 6] == 0, 5] = 0.0
end = T.max()
print(f"kernel span {end:.1f} us")
# ---- how many work-groups are between 'X loaded' and 'rows streamed' at a time
bins = np.arange(0, end + 100, 100.0)
busy = np.zeros(len(bins) - 1)
nitems = np.zeros(len(bins) - 1)
for k in range(len(bins) - 1):
    lo, hi = bins[k], bins[k + 1]
    ov = np.clip(np.minimum(T[:, 3], hi) - np.maximum(T[:, 2], lo), 0, None)
    busy[k] = ov.sum() / (hi - lo)
    nitems[k] = ((T[:, 1] >= lo) & (T[:, 1] < hi)).sum()
print("time bin (us) : work-groups streaming on average : items received in the bin")
for k in range(len(busy)):
    print(f"  {bins[k]:7.0f} : {busy[k]:7.1f} : {int(nitems[k])}")
# ---- phases of an average item, early and late
for name, sel in (("first third", T[:, 1] < end / 3), ("last third", T[:, 1] > 2 * end / 3)):
    if sel.sum() == 0:
        continue
    d = T[sel]
    print(f"{name}: {sel.sum()} items, rows/item {rows[sel].mean():.2f} | ticket->received {np.mean(d[:,1]-d[:,0]):.2f} "
          f"| X load {np.mean(d[:,2]-d[:,1]):.2f} | stream+sum+store {np.mean(d[:,3]-d[:,2]):.2f} "
          f"| arrive {np.mean(d[:,4]-d[:,3]):.2f} us")
    la = d[d[:, 5] > 0]
    if len(la):
        print(f"    last arrivers: {len(la)}, normalise+push {np.mean(la[:,5]-la[:,4]):.2f} us")
# ---- iteration chains of the graphs that finish last
fin = {}
for gg in np.unique(g):
    m = g == gg
    fin[gg] = T[m][:, 4].max()
late = sorted(fin, key=fin.get)[-4:]
for gg in late:
    m = np.where(g == gg)[0]
    m = m[np.argsort(T[m, 4])]
    chains, cur = [], []
    for idx in m:
        cur.append(idx)
        if T[idx, 5] > 0:
            chains.append(cur)
            cur = []
    print(f"graph {gg}: {len(chains)} iterations, finishes at {fin[gg]:.1f} us")
    prev_push = None
    rowsum = []
    for c in chains:
        c = np.array(c)
        got_first, got_last = T[c, 1].min(), T[c, 1].max()
        streamed_last, arrived_last, pushed = T[c, 3].max(), T[c, 4].max(), T[c, 5].max()
        if prev_push is not None:
            rowsum.append((got_first - prev_push, got_last - prev_push, np.mean(T[c, 2] - T[c, 1]),
                           np.mean(T[c, 3] - T[c, 2]), streamed_last - prev_push, arrived_last - streamed_last,
                           pushed - arrived_last, pushed - prev_push, len(c)))
        prev_push = pushed
    rs = np.array(rowsum)
    if len(rs) >= 10:
        for label, part in (("iterations 2-6", rs[:5]), ("last 10 iterations", rs[-10:])):
            a = part.mean(0)
            print(f"   {label}: push->first received {a[0]:.2f} | push->last received {a[1]:.2f} | X load {a[2]:.2f} "
                  f"| stream (mean item) {a[3]:.2f} | push->last streamed {a[4]:.2f} | ->arrived {a[5]:.2f} "
                  f"| normalise+push {a[6]:.2f} | whole iteration {a[7]:.2f} us | items {a[8]:.1f}")
# ---- streaming rate of a work-group by graph size (first third of the launch: the queue is backed up)
Rarr = np.array(Rh)[g]
sel0 = T[:, 1] < end / 3
shape = item & 0xF
print("size class : items : mean item KiB : stream us : GB/s per work-group while streaming : ticket->received : X load us")
for lo, hi in ((1, 16), (17, 24), (25, 32), (33, 40), (41, 48), (49, 56), (57, 64), (65, 80), (81, 128)):
    m = sel0 & (Rarr >= lo) & (Rarr <= hi)
    if m.sum() == 0:
        continue
    RS4 = (Rarr[m] + 3) // 4 * 4
    nblk = np.where(shape[m] <= 8, shape[m], 1)
    kib = rows[m] * Rarr[m] * Rarr[m] * RS4 * 4.0 / nblk / 1024.0
    st = T[m, 3] - T[m, 2]
    print(f"  {lo:3d}-{hi:3d} : {int(m.sum()):6d} : {kib.mean():9.1f} : {st.mean():7.2f} : {kib.sum() * 1024 / (st.sum() * 1e-6) / 1e9:7.2f} : "
          f"{np.mean(T[m, 1] - T[m, 0]):6.2f} : {np.mean(T[m, 2] - T[m, 1]):6.2f}")
