"""Lower bounds for the streamed MPM on the ragged c3 batch: total bytes over the HBM rate,
and the longest dependent chain of iterations of a single graph.
usage: python benchmarks/c3_bound.py"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "graph-vae_amd"))
import bench  # noqa: E402

sys.argv = [sys.argv[0], "--workload", "c3"]
args = bench.parse()
dev = torch.device("cuda:0")
model, batches, host, per_gpu, N, es = bench.make_model_and_batches(args, 0, 1, dev, "stream")
for b in batches:
    eps = torch.randn(per_gpu, 64, device=dev)
    model.forward_with_eps(b, eps)
    it = model.last_mpm_iterations().cpu().numpy().astype(np.float64)
    r = np.array(b.r_host, dtype=np.float64)
    per_iter = 4 * r ** 4 + 12 * r ** 2
    total = float((it * per_iter).sum())
    chain = it * np.maximum(per_iter / 7.0e12, 15e-6)
    k = int(np.argmax(chain))
    print(f"sizes: max {int(r.max())}, >64: {(r > 64).sum()}, mean {r.mean():.1f}; bytes {total/1e9:.2f} GB -> "
          f"{total/7.0e12*1e3:.2f} ms at 7 TB/s; longest chain: graph of {int(r[k])} nodes, {int(it[k])} iterations "
          f"-> {chain[k]*1e3:.2f} ms; top chains (ms): {np.sort(chain)[-4:][::-1].round(4)*1e3}")
    order = np.argsort(-r)
    print("  sizes (desc):", r[order][:20].astype(int).tolist())
    print("  iterations  :", it[order][:20].astype(int).tolist())
    alive_bytes = [float(per_iter[it > k].sum()) for k in range(50)]
    print("  bytes per round (MB) while graphs drop out:", [round(x / 1e6) for x in alive_bytes[::4]])
