"""Histogram of the MPM iterations per graph on the bench's c2 batches (early stop at 1e-4, at most
50): how many graphs run to the cap and how long the under-filled tail of the streamed MPM is.
usage: python benchmarks/iters_hist.py"""
import sys, torch, numpy as np
sys.path.insert(0, "graph-vae_amd")
from graphvae.data import synthetic_batch
from graphvae.models import GraphVAE
torch.manual_seed(42)
m = GraphVAE(1, 128, 64, 32, matching="stream")
host = [synthetic_batch("dense", 256, 32, "deg", seed=b) for b in range(2)]
m.precompute_dataset_rho(host); m = m.to("cuda:0").train()
for b in host:
    out = m(b.to("cuda:0"))
    it = m.last_mpm_iterations().cpu().numpy()
    print("iters: mean %.2f min %d max %d" % (it.mean(), it.min(), it.max()))
    print("sorted tail:", np.sort(it)[-24:])
    print("hist:", np.bincount(it, minlength=51)[:51])
