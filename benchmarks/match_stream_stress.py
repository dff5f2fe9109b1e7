"""Randomised check of gvae_match_stream (first MPM iteration from row maxima of S) against
gvae_edge_sim + gvae_mpm_stream: 12 seeds x 96 graphs of 1..48 nodes, 0/1 and weighted (signed)
adjacencies, probabilities scaled down to 1e-20 (denormal products) and up to 1e3, early stop on
and off.  Prints one line per case and the number of mismatches (expected 0).
usage: python benchmarks/match_stream_stress.py"""
import sys, numpy as np, torch
sys.path.insert(0, "graph-vae_amd")
from graphvae import _hip as hip
dev = torch.device("cuda:0")
bad = 0
for seed in range(12):
    rng = np.random.default_rng(100 + seed)
    sizes = [int(x) for x in rng.integers(1, 49, 96)]
    N = max(sizes)
    A = np.zeros((len(sizes), N, N), np.float32); P = np.zeros_like(A)
    for g, R in enumerate(sizes):
        scale = [1.0, 1e-3, 1e3, 1e-20][seed % 4]
        w = rng.normal(0.3, 1.0, (R, R)).astype(np.float32) * (rng.random((R, R)) < 0.4)
        if seed % 3 == 0: w = (w != 0).astype(np.float32)
        up = np.triu(w, 1); A[g, :R, :R] = up + up.T + np.eye(R, dtype=np.float32)
        lg = np.triu(rng.normal(0, 2.0, (R, R)).astype(np.float32), 1)
        P[g, :R, :R] = (1 / (1 + np.exp(-(lg + lg.T)))).astype(np.float32) * np.float32(scale if seed % 2 else 1.0)
    A = torch.from_numpy(A).to(dev); P = torch.from_numpy(P).to(dev)
    R = torch.tensor(sizes, dtype=torch.int32, device=dev)
    NS = hip.node_sim(A, P, R)
    for es in (True, False):
        S, S_off = hip.edge_sim(A, P, NS, R, sizes, hip.S_PADDED)
        Xs = hip.x0_uniform(sizes, N, dev)
        ds = hip.mpm_stream(S, S_off, R, N, Xs, 12, 1e-4, es, layout=hip.S_PADDED)
        Xm = hip.x0_uniform(sizes, N, dev)
        _, _, dm = hip.match_stream(A, P, NS, R, sizes, Xm, 12, 1e-4, es, layout=hip.S_PADDED)
        same = torch.equal(ds, dm) and all(torch.equal(Xm[g, :r, :r], Xs[g, :r, :r]) or
                                           (torch.isnan(Xm[g, :r, :r]).all() and torch.isnan(Xs[g, :r, :r]).all())
                                           for g, r in enumerate(sizes))
        bad += not same
        print(seed, es, "same" if same else "DIFFERENT", int(ds.min()), int(ds.max()))
print("mismatches:", bad)
