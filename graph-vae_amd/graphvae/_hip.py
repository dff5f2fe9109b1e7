"""ctypes binding of libgvae_hip.so (declared in include/gvae_hip.h).

The library is the product path: if it is missing or a call fails this module
raises; there is no CPU fallback.  torch is used only to own device memory and to
supply the current HIP stream.
"""
from __future__ import annotations

import ctypes
from ctypes import c_double, c_float, c_int, c_int32, c_int64, c_void_p, POINTER
from pathlib import Path

import torch

import os

# GVAE_HIP_LIB selects another build of the same library (kernel tuning experiments)
_LIB_PATH = Path(os.environ.get("GVAE_HIP_LIB") or
                 Path(__file__).resolve().parent / "lib" / "libgvae_hip.so")
_lib = None

MAX_NODES_LDS = 128


class GvaeHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise GvaeHipError(
                f"{_LIB_PATH} not found: build it with `make -C graph-vae_amd/csrc` "
                "(or __graft_entry__.build()); the GraphVAE HIP path has no fallback")
        L = ctypes.CDLL(str(_LIB_PATH))
        L.gvae_version.restype = c_int
        L.gvae_last_error.restype = ctypes.c_char_p
        L.gvae_profile_enable.argtypes = [c_int]
        L.gvae_profile_read.argtypes = [c_int, POINTER(c_double), POINTER(c_int64)]
        L.gvae_node_sim.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p,
                                    c_void_p]
        L.gvae_edge_sim.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p]
        L.gvae_mpm_stream_ws_bytes.argtypes = [c_int, c_int]
        L.gvae_mpm_stream_ws_bytes.restype = c_int64
        L.gvae_mpm_stream.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                      c_int, c_int64, c_int, c_float, c_int, c_void_p, c_void_p,
                                      c_void_p]
        L.gvae_mpm_stream_static.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                             c_int, c_int64, c_int, c_float, c_int, c_void_p, c_void_p]
        L.gvae_mpm_pool_fwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p]
        L.gvae_mpm_pool_bwd.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.gvae_set_stream_handoff_bytes.argtypes = [c_int64]
        L.gvae_mpm_stream_stats.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_int64), c_int, c_void_p]
        L.gvae_match_stream.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                        c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                        c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]
        L.gvae_mpm_fused.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                     c_int, c_void_p, c_int, c_int64, c_int, c_float, c_int,
                                     c_void_p, c_void_p]
        L.gvae_lsap.argtypes = [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_void_p, c_int,
                                c_void_p, c_void_p]
        L.gvae_mpm_fused_assign.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_float,
                                            c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise GvaeHipError(f"libgvae_hip error {rc}: {lib().gvae_last_error().decode()}")


def _stream() -> int:
    """The current stream of the CURRENT device: kernel launches go to the current device, so
    every tensor handed to the library must live there (`_dev` checks it)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _dev(t: torch.Tensor, dtype) -> torch.Tensor:
    if not t.is_cuda:
        raise GvaeHipError("expected a device tensor (the HIP path has no CPU fallback)")
    if t.device.index != torch._C._cuda_getDevice():
        raise GvaeHipError(f"tensor on {t.device} but the current device is cuda:"
                           f"{torch.cuda.current_device()}: wrap the call in torch.cuda.device(...)")
    if t.dtype != dtype or not t.is_contiguous():
        raise GvaeHipError(f"expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


# --------------------------------------------------------------------------- profiling
FAMILIES = {"edge_sim": 0, "mpm_stream": 1, "mpm_stream_iter": 1, "mpm_fused": 3, "lsap": 4,
            "node_sim": 5}


def profile_enable(on: bool) -> None:
    _check(lib().gvae_profile_enable(int(on)))


def profile_reset() -> None:
    _check(lib().gvae_profile_reset())


def profile_read(family: str):
    ms, n = c_double(0.0), c_int64(0)
    _check(lib().gvae_profile_read(FAMILIES[family], ctypes.byref(ms), ctypes.byref(n)))
    return ms.value, n.value


# --------------------------------------------------------------------------- matching
def node_sim(A: torch.Tensor, P: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
    """A, P: (G, N, N) fp32; R: (G,) int32.  Returns NS (G, N, N)."""
    _dev(A, torch.float32), _dev(P, torch.float32), _dev(R, torch.int32)
    G, N, _ = A.shape
    NS = torch.zeros_like(A)
    _check(lib().gvae_node_sim(A.data_ptr(), P.data_ptr(), R.data_ptr(), G, N, N * N,
                               NS.data_ptr(), _stream()))
    return NS


S_DENSE, S_PADDED = 0, 1


def s_block_floats(r: int, layout: int = S_DENSE) -> int:
    """Floats of one graph's block in the S buffer (a multiple of 4)."""
    r = int(r)
    rs = (r + 3) // 4 * 4 if layout == S_PADDED else r
    return (r ** 3 * rs + 3) // 4 * 4


def s_offsets(R_host, layout: int = S_DENSE):
    """Element offsets of each graph's block in a packed S buffer, and the total."""
    offs, tot = [], 0
    for r in R_host:
        offs.append(tot)
        tot += s_block_floats(r, layout)
    return offs, tot


def s_chunks(R_host, budget_floats: int, layout: int = S_PADDED):
    """Splits the batch into consecutive chunks whose S blocks fit `budget_floats` together.
    Returns (chunk_begin [n_chunks+1], offsets within the chunk [G], floats of the largest chunk)."""
    begin, offs, tot, largest = [0], [], 0, 0
    for g, r in enumerate(R_host):
        need = s_block_floats(r, layout)
        if need > budget_floats:
            raise GvaeHipError(f"one graph of {r} nodes needs {need * 4} bytes of S, over the "
                               f"budget of {budget_floats * 4} bytes")
        if tot + need > budget_floats:
            begin.append(g)
            tot = 0
        offs.append(tot)
        tot += need
        largest = max(largest, tot)
    begin.append(len(R_host))
    return begin, offs, largest


def edge_sim(A, P, NS, R, R_host, layout: int = S_DENSE) -> tuple[torch.Tensor, torch.Tensor]:
    _dev(A, torch.float32), _dev(P, torch.float32), _dev(NS, torch.float32), _dev(R, torch.int32)
    G, N, _ = A.shape
    offs, tot = s_offsets(R_host, layout)
    S = torch.empty(max(tot, 4), dtype=torch.float32, device=A.device)
    S_off = torch.tensor(offs, dtype=torch.int64, device=A.device)
    r_max = int(max(R_host)) if len(R_host) else 1
    _check(lib().gvae_edge_sim(A.data_ptr(), P.data_ptr(), NS.data_ptr(), R.data_ptr(), G, N, N * N,
                               max(r_max, 1), layout, S.data_ptr(), S_off.data_ptr(), _stream()))
    return S, S_off


def mpm_stream(S, S_off, R, r_max: int, X, iters: int = 50, tol: float = 1e-4,
               early_stop: bool = True, ws: torch.Tensor | None = None,
               layout: int = S_DENSE) -> torch.Tensor:
    """X: (G, N, N) fp32, X0 in / result out (leading R x R blocks).  Returns iters_done (G,) int32."""
    _dev(S, torch.float32), _dev(S_off, torch.int64), _dev(R, torch.int32), _dev(X, torch.float32)
    G, N, _ = X.shape
    need = lib().gvae_mpm_stream_ws_bytes(G, N)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=X.device)
    done = torch.empty(G, dtype=torch.int32, device=X.device)
    _check(lib().gvae_mpm_stream(S.data_ptr(), S_off.data_ptr(), layout, R.data_ptr(), G, r_max,
                                 X.data_ptr(), N, N * N, iters, tol, int(early_stop),
                                 done.data_ptr(), ws.data_ptr(), _stream()))
    return done


def set_stream_handoff_bytes(n: int) -> None:
    """Live similarity-tensor bytes below which the streamed matching of the step hands its
    last unconverged graphs to the S-free kernel (0: never; default 256 MiB)."""
    _check(lib().gvae_set_stream_handoff_bytes(int(n)))


def mpm_stream_stats(ws_ptr: int, reset: bool = False):
    """(algorithmic bytes, graph-iterations) the streaming kernel executed through the workspace
    at `ws_ptr` since the last reset; synchronises the current stream."""
    b, n = c_int64(0), c_int64(0)
    _check(lib().gvae_mpm_stream_stats(ws_ptr, ctypes.byref(b), ctypes.byref(n), int(reset), _stream()))
    return b.value, n.value


STATIC_SCRATCH_BYTES = 2 << 20


def mpm_stream_static(S, S_off, R, r_max: int, X, done, iters: int = 50, tol: float = 1e-4,
                      early_stop: bool = True, layout: int = S_DENSE) -> None:
    """gvae_mpm_stream_static: as `mpm_stream`, with the library's built-in scratch buffer (no
    workspace, nothing allocated by the call: `done` (G,) int32 comes from the caller too).
    Raises for batches whose workspace would exceed STATIC_SCRATCH_BYTES."""
    _dev(S, torch.float32), _dev(S_off, torch.int64), _dev(R, torch.int32), _dev(X, torch.float32)
    _dev(done, torch.int32)
    G, N, _ = X.shape
    if done.numel() != G:
        raise GvaeHipError("mpm_stream_static: `done` must have one entry per graph")
    _check(lib().gvae_mpm_stream_static(S.data_ptr(), S_off.data_ptr(), layout, R.data_ptr(), G, r_max,
                                        X.data_ptr(), N, N * N, iters, tol, int(early_stop),
                                        done.data_ptr(), _stream()))


def match_stream(A, P, NS, R, R_host, X, iters: int = 50, tol: float = 1e-4, early_stop: bool = True,
                 layout: int = S_PADDED, ws: torch.Tensor | None = None):
    """Similarity tensor + streamed MPM from the uniform start in one library call
    (gvae_match_stream): returns (S, S_off, iterations); X (G, N, N) is output only."""
    _dev(A, torch.float32), _dev(P, torch.float32), _dev(NS, torch.float32), _dev(R, torch.int32)
    _dev(X, torch.float32)
    G, N, _ = A.shape
    offs, total = s_offsets(R_host, layout)
    S = torch.empty(max(total, 4), dtype=torch.float32, device=A.device)
    S_off = torch.tensor(offs, dtype=torch.int64, device=A.device)
    done = torch.empty(G, dtype=torch.int32, device=A.device)
    need = lib().gvae_mpm_stream_ws_bytes(G, X.shape[1])
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=A.device)
    _check(lib().gvae_match_stream(A.data_ptr(), P.data_ptr(), NS.data_ptr(), R.data_ptr(), G, N, N * N,
                                   max(R_host), layout, S.data_ptr(), S_off.data_ptr(), X.data_ptr(),
                                   X.shape[1], X.shape[1] * X.shape[2], iters, tol, int(early_stop),
                                   done.data_ptr(), ws.data_ptr(), _stream()))
    return S, S_off, done


def mpm_pool_fwd(S, X) -> torch.Tensor:
    """Unnormalised update of one MPM iteration for a dense (R,R,R,R) S and an (R,R) X."""
    _dev(S, torch.float32), _dev(X, torch.float32)
    R = X.shape[0]
    Y = torch.empty_like(X)
    _check(lib().gvae_mpm_pool_fwd(S.data_ptr(), R, X.data_ptr(), Y.data_ptr(), _stream()))
    return Y


def mpm_pool_bwd(S, X, gY, gS=None) -> torch.Tensor:
    """Gradient of mpm_pool_fwd w.r.t. X (returned); w.r.t. S added onto gS when given."""
    _dev(S, torch.float32), _dev(X, torch.float32), _dev(gY, torch.float32)
    R = X.shape[0]
    gX = torch.empty_like(X)
    _check(lib().gvae_mpm_pool_bwd(S.data_ptr(), R, X.data_ptr(), gY.data_ptr(), gX.data_ptr(),
                                   gS.data_ptr() if gS is not None else None, _stream()))
    return gX


def mpm_fused_supported(r_max: int) -> bool:
    """Whether the S-free MPM (one work-group per graph, the graph resident in LDS) takes graphs
    of this size: up to GVAE_MAX_NODES_LDS nodes."""
    return 0 < r_max <= MAX_NODES_LDS


def mpm_fused(A, P, NS, R, r_max: int, X, iters: int = 50, tol: float = 1e-4,
              early_stop: bool = True) -> torch.Tensor:
    _dev(A, torch.float32), _dev(P, torch.float32), _dev(NS, torch.float32), _dev(R, torch.int32)
    _dev(X, torch.float32)
    G, N, _ = A.shape
    done = torch.empty(G, dtype=torch.int32, device=X.device)
    _check(lib().gvae_mpm_fused(A.data_ptr(), P.data_ptr(), NS.data_ptr(), R.data_ptr(), G, N, N * N,
                                r_max, X.data_ptr(), X.shape[1], X.shape[1] * X.shape[2], iters, tol,
                                int(early_stop), done.data_ptr(), _stream()))
    return done


def mpm_fused_assign(A, P, NS, R, r_max: int, X, iters: int = 50, tol: float = 1e-4,
                     early_stop: bool = True, n_perm: int | None = None):
    """S-free MPM with the assignment as the kernel's epilogue: returns (iterations (G,) int32,
    perm (G, n_perm) int64, status (G,) int32); X is updated in place."""
    _dev(A, torch.float32), _dev(P, torch.float32), _dev(NS, torch.float32), _dev(R, torch.int32)
    _dev(X, torch.float32)
    G, N, _ = A.shape
    n_perm = N if n_perm is None else n_perm
    done = torch.empty(G, dtype=torch.int32, device=X.device)
    perm = torch.empty(G, n_perm, dtype=torch.int64, device=X.device)
    status = torch.empty(G, dtype=torch.int32, device=X.device)
    _check(lib().gvae_mpm_fused_assign(A.data_ptr(), P.data_ptr(), NS.data_ptr(), R.data_ptr(), G, N,
                                       N * N, r_max, X.data_ptr(), X.shape[1],
                                       X.shape[1] * X.shape[2], iters, tol, int(early_stop),
                                       done.data_ptr(), perm.data_ptr(), n_perm, status.data_ptr(),
                                       _stream()))
    return done, perm, status


def lsap(X, R, r_max: int, n_perm: int | None = None):
    """Returns (perm (G, n_perm) int64, status (G,) int32)."""
    _dev(X, torch.float32), _dev(R, torch.int32)
    G, N, _ = X.shape
    n_perm = N if n_perm is None else n_perm
    perm = torch.empty(G, n_perm, dtype=torch.int64, device=X.device)
    status = torch.empty(G, dtype=torch.int32, device=X.device)
    _check(lib().gvae_lsap(X.data_ptr(), N, N * N, R.data_ptr(), G, r_max, perm.data_ptr(), n_perm,
                           status.data_ptr(), _stream()))
    return perm, status


def x0_uniform(R_host, N: int, device) -> torch.Tensor:
    """X0 = full((R,R), 1/R) per graph (reference model.py:317), zero elsewhere."""
    X = torch.zeros(len(R_host), N, N, dtype=torch.float32, device=device)
    for g, r in enumerate(R_host):
        X[g, :r, :r] = 1.0 / r
    return X
