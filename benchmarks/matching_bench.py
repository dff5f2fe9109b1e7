"""Micro-benchmark of the matching kernels (device-resident inputs).
usage: python benchmarks/matching_bench.py [N] [B] [iters] [early_stop 0|1] [padded S rows 0|1]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae import _hip  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    es = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    layout = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    up = torch.triu((torch.rand(B, N, N, generator=gen) < 0.5).float(), 1)
    A = (up + up.transpose(1, 2) + torch.eye(N)).to(dev)
    lg = torch.triu(torch.randn(B, N, N, generator=gen) * 0.3, 1)
    P = torch.sigmoid(lg + lg.transpose(1, 2)).to(dev).contiguous()
    Rh = [N] * B
    R = torch.tensor(Rh, dtype=torch.int32, device=dev)
    NS = _hip.node_sim(A, P, R)
    S, S_off = _hip.edge_sim(A, P, NS, R, Rh, layout)
    torch.cuda.synchronize()
    sbytes = 4.0 * B * N ** 4

    def timeit(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    t = timeit(lambda: _hip.edge_sim(A, P, NS, R, Rh, layout))
    print(f"edge_sim   N={N} B={B}: {t*1e3:8.3f} ms  {sbytes/t/1e12:6.3f} TB/s (write, incl. alloc)")
    ws = torch.empty(_hip.lib().gvae_mpm_stream_ws_bytes(B, N), dtype=torch.uint8, device=dev)
    X0 = _hip.x0_uniform(Rh, N, dev)

    def run_stream():
        X = X0.clone()
        return _hip.mpm_stream(S, S_off, R, N, X, iters, 1e-4, es, ws=ws, layout=layout)

    _hip.profile_enable(True)
    _hip.profile_reset()
    done = run_stream()
    torch.cuda.synchronize()
    ms_it, n_it = _hip.profile_read("mpm_stream_iter")
    ms_nm, n_nm = 0.0, 0
    _hip.profile_enable(False)
    _hip.profile_reset()
    tot_it = float(done.sum())
    bytes_total = tot_it * (4.0 * N ** 4 + 12.0 * N * N)
    print(f"mpm_stream N={N} B={B} iters={iters} es={es}: iter kernels {ms_it:8.3f} ms over {n_it} launches, "
          f"norm {ms_nm:7.3f} ms; executed graph-iters {tot_it:.0f}; "
          f"{bytes_total/ (ms_it*1e-3)/1e12:6.3f} TB/s algorithmic")
    t = timeit(run_stream)
    print(f"mpm_stream wall {t*1e3:8.3f} ms  -> {B/t:9.1f} graphs/s (matching only)")

    if _hip.mpm_fused_supported(N):
        def run_fused():
            X = X0.clone()
            return _hip.mpm_fused(A, P, NS, R, N, X, iters, 1e-4, es)
        t = timeit(run_fused)
        print(f"mpm_fused  wall {t*1e3:8.3f} ms  -> {B/t:9.1f} graphs/s")
    Xf = X0.clone()
    _hip.mpm_stream(S, S_off, R, N, Xf, iters, 1e-4, es, ws=ws, layout=layout)
    t = timeit(lambda: _hip.lsap(Xf, R, N))
    print(f"lsap       wall {t*1e3:8.3f} ms")


if __name__ == "__main__":
    main()
