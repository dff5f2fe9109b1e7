"""gvae_mpm_stream_static (the entry point that allocates nothing: its workspace is a built-in
device array) against gvae_mpm_stream on the same inputs, through the C ABI: X and iteration
counts bit for bit, on two calls in a row, the second with a smaller input (the array keeps the
first call's contents), then the size limit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(Rh, N, seed):
    from graphvae import _hip
    rng = np.random.default_rng(seed)
    A = np.zeros((len(Rh), N, N), np.float32)
    P = np.zeros_like(A)
    for g, r in enumerate(Rh):
        up = np.triu((rng.random((r, r)) < 0.3).astype(np.float32), 1)
        A[g, :r, :r] = up + up.T + np.eye(r, dtype=np.float32)
        lg = np.triu(rng.normal(0, 1, (r, r)), 1)
        P[g, :r, :r] = 1 / (1 + np.exp(-(lg + lg.T)))
    Ad, Pd = torch.from_numpy(A).to(DEV), torch.from_numpy(P).to(DEV)
    R = torch.tensor(Rh, dtype=torch.int32, device=DEV)
    NS = _hip.node_sim(Ad, Pd, R)
    S, S_off = _hip.edge_sim(Ad, Pd, NS, R, Rh)
    return S, S_off, R


def test_static_scratch_entry_point_equals_workspace_entry_point():
    from graphvae import _hip
    rng = np.random.default_rng(0)
    cases = [([int(r) for r in rng.integers(2, 33, 48)], 32, 50, True),    # first call
             ([9, 16, 3, 12, 7], 16, 20, False),                           # second, smaller
             ([104, 77, 20, 5], 104, 6, False)]    # 512-thread work-groups, row maxima bounded in LDS
    for k, (Rh, N, iters, early) in enumerate(cases):
        S, S_off, R = _problem(Rh, N, seed=10 + k)
        X_ws = _hip.x0_uniform(Rh, N, DEV)
        done_ws = _hip.mpm_stream(S, S_off, R, max(Rh), X_ws, iters, 1e-4, early)
        X_st = _hip.x0_uniform(Rh, N, DEV)
        done_st = torch.full((len(Rh),), -7, dtype=torch.int32, device=DEV)
        before = torch.cuda.memory_allocated()
        _hip.mpm_stream_static(S, S_off, R, max(Rh), X_st, done_st, iters, 1e-4, early)
        assert torch.cuda.memory_allocated() == before        # nothing taken from torch either
        torch.cuda.synchronize()
        assert torch.equal(done_st, done_ws), k
        assert int(done_st.min()) >= 1
        assert torch.equal(X_st, X_ws), k
    # beyond the built-in buffer: a clear error, no launch
    G, N = 400, 32
    assert _hip.lib().gvae_mpm_stream_ws_bytes(G, N) > _hip.STATIC_SCRATCH_BYTES
    X = torch.zeros(G, N, N, device=DEV)
    with pytest.raises(_hip.GvaeHipError, match="scratch"):
        _hip.mpm_stream_static(S, torch.zeros(G, dtype=torch.int64, device=DEV),
                               torch.full((G,), 4, dtype=torch.int32, device=DEV), N, X,
                               torch.zeros(G, dtype=torch.int32, device=DEV))
