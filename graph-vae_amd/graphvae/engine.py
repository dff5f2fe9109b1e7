"""Host-side sequencing of the HIP training step (ctypes calls into libgvae_hip.so).

torch owns the device buffers and the stream; every computation is a HIP kernel
behind the C ABI of include/gvae_hip.h.  There is no CPU or eager fallback.
"""
from __future__ import annotations

import contextlib
import ctypes
from ctypes import POINTER, c_float, c_int, c_int32, c_int64, c_void_p

import torch

from . import _hip
from .data import DenseGraphBatch

P_COUNT = 20
MATCH_NONE, MATCH_STREAM, MATCH_FUSED = 0, 1, 2
_MATCH = {"none": MATCH_NONE, "stream": MATCH_STREAM, "fused": MATCH_FUSED}
LOSS_CURRENT, LOSS_DUMMY_MASKED, LOSS_INDIVIDUAL_RHO = 0, 1, 2   # GVAE_LOSS_* of the library
# GVAE_PHASE_* of the library: the parts of the forward's second half, in their order
PHASE_PREP, PHASE_MPM, PHASE_TAIL, PHASE_ASSIGN_LOSS = 4, 8, 16, 2
PHASE_MATCH = PHASE_PREP | PHASE_MPM | PHASE_TAIL
_PHASE_ORDER = (PHASE_PREP, PHASE_MPM, PHASE_TAIL, PHASE_ASSIGN_LOSS)

# parameter tensors in flat-buffer order (reference module named_parameters() order)
PARAM_NAMES = [
    "conv1.bias", "conv1.lin.weight", "bn1.weight", "bn1.bias",
    "conv2.bias", "conv2.lin.weight", "bn2.weight", "bn2.bias",
    "mu.weight", "mu.bias", "logvar.weight", "logvar.bias",
    "dec_edge.0.weight", "dec_edge.0.bias", "dec_edge.2.weight", "dec_edge.2.bias",
    "dec_node.0.weight", "dec_node.0.bias", "dec_node.2.weight", "dec_node.2.bias",
]


class Dims(ctypes.Structure):
    _fields_ = [("F", c_int32), ("H", c_int32), ("Z", c_int32), ("N", c_int32)]


_bound = False


def _bind():
    global _bound
    L = _hip.lib()
    if _bound:
        return L
    L.gvae_param_layout.argtypes = [POINTER(Dims), POINTER(c_int64), POINTER(c_int64)]
    L.gvae_param_layout.restype = c_int64
    L.gvae_step_ws_bytes.argtypes = [POINTER(Dims), c_int]
    L.gvae_step_ws_bytes.restype = c_int64
    L.gvae_step_forward.argtypes = [POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_float, c_int, c_void_p, c_void_p, POINTER(c_int32), c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.gvae_step_backward.argtypes = [POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p,
                                     c_void_p, c_void_p, c_void_p]
    L.gvae_step_forward_model.argtypes = [POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p]
    L.gvae_step_forward_match_loss.argtypes = [POINTER(Dims), c_void_p, c_void_p, c_int, c_int,
                                               c_void_p, c_int, c_int, c_int, c_float, c_int,
                                               c_void_p, c_void_p, POINTER(c_int32), c_int, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    L.gvae_step_read_debug.argtypes = [POINTER(Dims), c_int, c_void_p, c_int, POINTER(c_void_p),
                                       POINTER(c_int64)]
    L.gvae_adam_ws_bytes.restype = c_int64
    L.gvae_step_accumulate_loss.argtypes = [c_void_p, c_float, c_void_p, c_void_p, c_void_p]
    L.gvae_adam_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float,
                                 c_float, c_float, c_float, c_float, c_float, c_int64, c_void_p,
                                 c_void_p, c_void_p]
    L.gvae_sample_ws_bytes.argtypes = [POINTER(Dims), c_int]
    L.gvae_sample_ws_bytes.restype = c_int64
    L.gvae_sample_decode.argtypes = [POINTER(Dims), c_void_p, c_void_p, c_int, c_float, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]
    _bound = True
    return L


def sample_decode(F: int, H: int, Z: int, N: int, flat_params: torch.Tensor, z: torch.Tensor,
                  th_edge: float = 0.5, th_node: float = 0.5) -> dict:
    """Decoder pass of the sampler on the GPU (gvae_sample_decode): z (n, Z) -> edge logits
    (n, N(N-1)/2), node logits (n, N), adjacency bits (n, N, N) uint8, node-existence bits
    (n, N) uint8 and the number of existing nodes per sample (n,) int32."""
    L = _bind()
    if not (flat_params.is_cuda and z.is_cuda):
        raise _hip.GvaeHipError("sample_decode needs device tensors (no CPU fallback)")
    if z.dim() != 2 or z.shape[1] != Z or z.dtype != torch.float32:
        raise ValueError(f"z must be (n, {Z}) float32, got {tuple(z.shape)} {z.dtype}")
    z = z.contiguous()
    n, dev, d = z.shape[0], z.device, Dims(F, H, Z, N)
    edge = torch.empty(n, N * (N - 1) // 2, dtype=torch.float32, device=dev)
    node = torch.empty(n, N, dtype=torch.float32, device=dev)
    adj = torch.empty(n, N, N, dtype=torch.uint8, device=dev)
    exists = torch.empty(n, N, dtype=torch.uint8, device=dev)
    count = torch.empty(n, dtype=torch.int32, device=dev)
    ws = torch.empty(L.gvae_sample_ws_bytes(ctypes.byref(d), n), dtype=torch.uint8, device=dev)
    _hip._check(L.gvae_sample_decode(ctypes.byref(d), flat_params.data_ptr(), z.data_ptr(), n,
                                     float(th_edge), float(th_node), edge.data_ptr(), node.data_ptr(),
                                     adj.data_ptr(), exists.data_ptr(), count.data_ptr(),
                                     ws.data_ptr(), _hip._stream()))
    return {"edge_logits": edge, "node_logits": node, "adj": adj, "node_exists": exists,
            "num_nodes": count}


def param_layout(F: int, H: int, Z: int, N: int):
    """(offsets, sizes, total) of the flat parameter buffer, in floats."""
    L = _bind()
    d = Dims(F, H, Z, N)
    off = (c_int64 * P_COUNT)()
    size = (c_int64 * P_COUNT)()
    tot = L.gvae_param_layout(ctypes.byref(d), off, size)
    if tot < 0:
        raise _hip.GvaeHipError(f"bad model dimensions F={F} H={H} Z={Z} N={N}")
    return list(off), list(size), int(tot)


class _Slot:
    """Everything one step in flight owns: workspace (activations, logits, X), permutations,
    iteration counts, the four output scalars and the batch they belong to."""

    def __init__(self, device) -> None:
        self.ws = None
        self.ws_graphs = 0
        self.out4 = torch.zeros(4, dtype=torch.float32, device=device)
        self.perm = None
        self.iters_done = None
        self.fwd_batch = None
        self.pending = None


def _slot_attr(name):
    return property(lambda self: getattr(self._slots[self._cur], name),
                    lambda self, v: setattr(self._slots[self._cur], name, v))


class StepEngine:
    """Runs forward / backward / optimiser of one GraphVAE step for batches of up to any
    size; scratch is grown on demand and reused.  A caller that keeps several steps in flight
    (the matching half of one step beside the model half of the next ones) rotates over as many
    slots with `use_slot`; `perm`, `iters_done`, `out4` and the workspace are those of the
    current slot.  The similarity-tensor buffer is shared: matchings run one after the other."""

    _ws = _slot_attr("ws")
    _ws_graphs = _slot_attr("ws_graphs")
    out4 = _slot_attr("out4")
    perm = _slot_attr("perm")
    iters_done = _slot_attr("iters_done")
    _fwd_batch = _slot_attr("fwd_batch")
    _pending = _slot_attr("pending")

    def __init__(self, F: int, H: int, Z: int, N: int, device, matching: str = "auto",
                 mpm_iters: int = 50, tol: float = 1e-4, early_stop: bool = True,
                 s_budget_bytes: int | None = None):
        self.L = _bind()
        self.dims = Dims(F, H, Z, N)
        self.F, self.H, self.Z, self.N = F, H, Z, N
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _hip.GvaeHipError("the GraphVAE HIP step needs a GPU device (no CPU fallback)")
        if matching not in ("auto", "none", "stream", "fused"):
            raise ValueError("matching must be auto | none | stream | fused")
        self.matching = matching
        self.mpm_iters, self.tol, self.early_stop = int(mpm_iters), float(tol), bool(early_stop)
        self._slots = [_Slot(self.device)]
        self._cur = 0
        # the streamed matching keeps the similarity tensor of as many graphs as fit this many
        # bytes and walks the batch in chunks (default: a third of the device's memory)
        if s_budget_bytes is None:
            s_budget_bytes = torch.cuda.get_device_properties(self.device).total_memory // 3
        self.s_budget_bytes = int(s_budget_bytes)
        self._S = None
        self._S_off = None
        self._S_key = None
        self._S_chunks = None
        self._adam_ws = torch.empty(self.L.gvae_adam_ws_bytes(), dtype=torch.uint8, device=self.device)
        self.generation = 0
        self.last_mode = MATCH_NONE

    def use_slot(self, k: int) -> None:
        """Makes slot k (created on first use) the one the following calls work on."""
        while len(self._slots) <= k:
            self._slots.append(_Slot(self.device))
        self._cur = k

    # ------------------------------------------------------------------ helpers
    def _ensure(self, B: int) -> None:
        if self._ws is None or B > self._ws_graphs:
            need = self.L.gvae_step_ws_bytes(ctypes.byref(self.dims), B)
            # (zeroed once: the streamed matching's statistics words in it are only ever added to)
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self._ws_graphs = B
        if self.perm is None or self.perm.shape[0] != B:
            self.perm = torch.empty(B, self.N, dtype=torch.int64, device=self.device)
            self.iters_done = torch.empty(B, dtype=torch.int32, device=self.device)

    def resolve_matching(self, r_max: int) -> int:
        if self.matching == "auto":
            return MATCH_FUSED if _hip.mpm_fused_supported(r_max) else MATCH_STREAM
        return _MATCH[self.matching]

    def _ensure_S(self, r_host, stream=None) -> None:
        key = tuple(r_host)
        if key == self._S_key:
            return
        begin, offs, largest = _hip.s_chunks(r_host, self.s_budget_bytes // 4)
        # allocation and upload on the stream the matching runs on: the kernels that read the
        # offsets are ordered behind the copy only there
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            if self._S is None or self._S.numel() < largest:
                self._S = None  # release before growing
                self._S = torch.empty(max(largest, 4), dtype=torch.float32, device=self.device)
            # staged through pinned memory: a pageable upload would make the host wait for the
            # device to drain on every batch with a new size pattern
            self._S_off = torch.tensor(offs, dtype=torch.int64).pin_memory().to(self.device, non_blocking=True)
        self._S_chunks = (c_int32 * len(begin))(*begin)
        self._S_key = key

    def _check_batch(self, b: DenseGraphBatch) -> None:
        B, N, F = b.x.shape
        if N != self.N or F != self.F or tuple(b.adj_dense.shape) != (B, N, N):
            raise _hip.GvaeHipError(
                f"batch shapes x={tuple(b.x.shape)} adj={tuple(b.adj_dense.shape)} do not match the "
                f"model (max_nodes={self.N}, in_dim={self.F})")
        for t, dt in ((b.x, torch.float32), (b.adj_dense, torch.float32), (b.n_x, torch.int32),
                      (b.num_real_nodes, torch.int32)):
            _hip._dev(t, dt)
        if len(b.r_host) != B or min(b.r_host) < 1 or max(b.r_host) > N:
            raise _hip.GvaeHipError("batch.r_host must list 1 <= R <= max_nodes for every graph")

    # ------------------------------------------------------------------ step pieces
    def _prepare(self, params, bn_state, batch, eps, pos_weight, loss_variant):
        self._check_batch(batch)
        B = batch.num_graphs
        self._ensure(B)
        _hip._dev(params, torch.float32), _hip._dev(bn_state, torch.float32)
        _hip._dev(eps, torch.float32), _hip._dev(pos_weight, torch.float32)
        if tuple(eps.shape) != (B, self.Z):
            raise _hip.GvaeHipError(f"eps must be ({B}, {self.Z})")
        if loss_variant not in (LOSS_CURRENT, LOSS_DUMMY_MASKED, LOSS_INDIVIDUAL_RHO):
            raise ValueError(f"unknown loss variant {loss_variant}")
        r_max = max(self._matched_sizes(batch, loss_variant))
        mode = self.resolve_matching(r_max)
        if loss_variant == LOSS_INDIVIDUAL_RHO and mode == MATCH_NONE:
            raise _hip.GvaeHipError("the individual-rho loss needs a matching (its loss depends on the permutation)")
        return B, r_max, mode

    def _matched_sizes(self, batch: DenseGraphBatch, loss_variant: int):
        """Node counts the matching runs over: the real nodes, or all max_nodes rows for the
        individual-rho loss (reference model_legacy.py:294-300)."""
        return (self.N,) * batch.num_graphs if loss_variant == LOSS_INDIVIDUAL_RHO else batch.r_host

    def _S_args(self, mode: int, batch: DenseGraphBatch, loss_variant: int = LOSS_CURRENT, stream=None):
        if mode != MATCH_STREAM:
            return None, None, None, 0
        self._ensure_S(self._matched_sizes(batch, loss_variant), stream)
        # the shared buffers are used on whichever stream runs the matching: the allocator must
        # not hand their memory out again while that stream still reads it
        cur = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._S.record_stream(cur)
        self._S_off.record_stream(cur)
        return (self._S.data_ptr(), self._S_off.data_ptr(), self._S_chunks, len(self._S_chunks) - 1)

    def forward(self, params: torch.Tensor, bn_state: torch.Tensor, batch: DenseGraphBatch,
                eps: torch.Tensor, pos_weight: torch.Tensor, train: bool,
                loss_variant: int = LOSS_CURRENT) -> torch.Tensor:
        """Runs the forward; returns the device tensor [rec, kl, max_logit, logvar_max]
        (a view of engine memory, overwritten by the next forward).  `loss_variant`:
        LOSS_CURRENT (reference model.py:281-373), LOSS_DUMMY_MASKED (`forward_old`,
        model.py:378-440) or LOSS_INDIVIDUAL_RHO (`forward_individual_rho`,
        model_legacy.py:257-363); the backward of this batch uses the same variant."""
        B, r_max, mode = self._prepare(params, bn_state, batch, eps, pos_weight, loss_variant)
        S_ptr, Soff_ptr, chunks, n_chunks = self._S_args(mode, batch, loss_variant)
        _hip._check(self.L.gvae_step_forward(
            ctypes.byref(self.dims), params.data_ptr(), bn_state.data_ptr(), batch.x.data_ptr(),
            batch.adj_dense.data_ptr(), batch.n_x.data_ptr(), batch.num_real_nodes.data_ptr(), B, r_max,
            eps.data_ptr(), pos_weight.data_ptr(), int(loss_variant), int(train), mode, self.mpm_iters,
            self.tol, int(self.early_stop), S_ptr, Soff_ptr, chunks, n_chunks, self._ws.data_ptr(),
            self.out4.data_ptr(),
            self.perm.data_ptr(), self.iters_done.data_ptr(), _hip._stream()))
        self.generation += 1
        self._fwd_batch = (batch, eps, int(loss_variant))
        self.last_mode = mode
        return self.out4

    def forward_model(self, params: torch.Tensor, bn_state: torch.Tensor, batch: DenseGraphBatch,
                      eps: torch.Tensor, pos_weight: torch.Tensor, train: bool,
                      loss_variant: int = LOSS_CURRENT) -> None:
        """First half of `forward` (encoder, heads, decoder) on the current stream: all that
        `backward` needs.  `forward_match_loss` must follow for the same batch, on this or on
        another stream ordered behind this call."""
        if loss_variant == LOSS_INDIVIDUAL_RHO:
            raise _hip.GvaeHipError("the individual-rho loss has no split forward: its backward reads the "
                                    "permutation, so the whole forward() has to precede it")
        B, r_max, mode = self._prepare(params, bn_state, batch, eps, pos_weight, loss_variant)
        _hip._check(self.L.gvae_step_forward_model(
            ctypes.byref(self.dims), params.data_ptr(), bn_state.data_ptr(), batch.x.data_ptr(),
            batch.adj_dense.data_ptr(), batch.n_x.data_ptr(), batch.num_real_nodes.data_ptr(), B,
            eps.data_ptr(), int(train), self._ws.data_ptr(), _hip._stream()))
        self.generation += 1
        self._fwd_batch = (batch, eps, int(loss_variant))
        self._pending = (B, r_max, mode, pos_weight, PHASE_MATCH | PHASE_ASSIGN_LOSS)
        self.last_mode = mode

    def forward_match_loss(self, phases: int = PHASE_MATCH | PHASE_ASSIGN_LOSS,
                           stream: "torch.cuda.Stream | None" = None) -> torch.Tensor:
        """Second half of `forward` (matching, assignment, losses) for the batch of the last
        `forward_model` on this slot, enqueued on `stream` (default: the current one; passing the
        stream saves the caller a stream context per call); returns the device tensor
        [rec, kl, max_logit, logvar_max] (engine memory, valid once PHASE_ASSIGN_LOSS has run).
        The parameter gradients do not depend on the permutation (reference model.py:325-334
        gathers prediction and truth through the same permutation), so this may run concurrently
        with `backward`.  `phases`: the parts PHASE_PREP (probabilities, node block, similarity
        tensor, first iteration), PHASE_MPM (the remaining iterations), PHASE_TAIL (streamed
        matching: the iterations left to the S-free kernel) and PHASE_ASSIGN_LOSS may be enqueued
        by several calls, in this order, each ordered behind the one before (possibly on another
        stream); PHASE_MATCH is the first three."""
        if self._pending is None:
            raise _hip.GvaeHipError("forward_match_loss called without a forward_model before it")
        B, r_max, mode, pos_weight, todo = self._pending
        want = [p for p in _PHASE_ORDER if phases & p]
        if not want or phases & ~todo or [p for p in _PHASE_ORDER if todo & p][:len(want)] != want:
            raise _hip.GvaeHipError(f"forward_match_loss: phases {phases} are not the next outstanding "
                                    f"ones ({todo}; order: prep 4, mpm 8, tail 16, assign + loss 2)")
        batch, _eps, loss_variant = self._fwd_batch
        todo &= ~phases
        self._pending = (B, r_max, mode, pos_weight, todo) if todo else None
        # (the first part plans the chunks and uploads their offsets, on its stream: the later
        # parts are ordered behind it)
        touches_S = phases & (PHASE_PREP | PHASE_MPM)
        S_ptr, Soff_ptr, chunks, n_chunks = (self._S_args(mode, batch, stream=stream) if touches_S
                                             else (None, None, None, 0))
        if mode == MATCH_STREAM and not touches_S:
            # (the buffer is validated, not used; the chunk list tells the later parts whether
            # the matching ran as one chunk, i.e. left its S-free tail to PHASE_TAIL)
            S_ptr, Soff_ptr = self._S.data_ptr(), self._S_off.data_ptr()
            chunks, n_chunks = self._S_chunks, len(self._S_chunks) - 1
        _hip._check(self.L.gvae_step_forward_match_loss(
            ctypes.byref(self.dims), batch.adj_dense.data_ptr(), batch.num_real_nodes.data_ptr(), B,
            r_max, pos_weight.data_ptr(), loss_variant, mode, self.mpm_iters, self.tol,
            int(self.early_stop), S_ptr, Soff_ptr, chunks, n_chunks, self._ws.data_ptr(),
            self.out4.data_ptr(), self.perm.data_ptr(), self.iters_done.data_ptr(), int(phases),
            stream.cuda_stream if stream is not None else _hip._stream()))
        return self.out4

    def accumulate_loss(self, out4: torch.Tensor, beta: float, loss_sum: torch.Tensor,
                        copy: "torch.Tensor | None" = None, stream: "torch.cuda.Stream | None" = None) -> None:
        """loss_sum[0] (float64) += out4[0] + beta * out4[1], and `copy` (4 floats) = out4, as one
        launch on `stream` (default: the current one)."""
        _hip._dev(out4, torch.float32), _hip._dev(loss_sum, torch.float64)
        if copy is not None:
            _hip._dev(copy, torch.float32)
        _hip._check(self.L.gvae_step_accumulate_loss(
            out4.data_ptr(), float(beta), copy.data_ptr() if copy is not None else None,
            loss_sum.data_ptr(), stream.cuda_stream if stream is not None else _hip._stream()))

    def backward(self, params: torch.Tensor, pos_weight: torch.Tensor, grads: torch.Tensor,
                 c_rec: float = 1.0, c_kl: float = 1.0, d_up: torch.Tensor | None = None) -> None:
        """Gradients of c_rec*up[0]*rec + c_kl*up[1]*kl into the flat `grads` buffer, for the
        batch of the most recent forward."""
        if self._fwd_batch is None:
            raise _hip.GvaeHipError("backward called before forward")
        batch, eps, loss_variant = self._fwd_batch
        _hip._dev(grads, torch.float32)
        up = None
        if d_up is not None:
            _hip._dev(d_up, torch.float32)
            up = d_up.data_ptr()
        _hip._check(self.L.gvae_step_backward(
            ctypes.byref(self.dims), params.data_ptr(), batch.x.data_ptr(), batch.adj_dense.data_ptr(),
            batch.n_x.data_ptr(), batch.num_real_nodes.data_ptr(), batch.num_graphs, eps.data_ptr(),
            pos_weight.data_ptr(), loss_variant, float(c_rec), float(c_kl), up, self._ws.data_ptr(),
            grads.data_ptr(), _hip._stream()))

    def adam_step(self, params, grads, m, v, step: int, lr: float, betas=(0.9, 0.999),
                  eps: float = 1e-8, weight_decay: float = 0.0, clip_max: float = 0.0,
                  grad_scale: float = 1.0, norm_out: torch.Tensor | None = None) -> None:
        n = params.numel()
        for t in (params, grads, m, v):
            _hip._dev(t, torch.float32)
            if t.numel() != n:
                raise _hip.GvaeHipError("adam_step: flat buffers differ in length")
        _hip._check(self.L.gvae_adam_step(
            params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), n, float(grad_scale),
            float(clip_max), float(lr), float(betas[0]), float(betas[1]), float(eps),
            float(weight_decay), int(step), self._adam_ws.data_ptr(),
            norm_out.data_ptr() if norm_out is not None else None, _hip._stream()))

    def check_matching(self) -> None:
        """Synchronises and raises if the last forward of any slot reported a fault in its
        matching (a negative iteration count from the streamed MPM's work queue, or a
        non-finite assignment cost)."""
        cur = self._cur
        try:
            for k in range(len(self._slots)):
                self._cur = k
                if self.iters_done is None:
                    continue
                if bool((self.iters_done < 0).any()):
                    raise _hip.GvaeHipError("streamed MPM: the device work queue faulted; X is invalid")
                B = self.iters_done.shape[0]
                if self.last_mode != MATCH_NONE and bool((self.debug_tensor("status", B) != 0).any()):
                    raise _hip.GvaeHipError("assignment: non-finite or infeasible cost matrix")
        finally:
            self._cur = cur

    def stream_stats(self, reset: bool = False):
        """(algorithmic bytes, graph-iterations) the streaming MPM kernel has executed for the
        current slot's batch size since the last reset (gvae_mpm_stream_stats); synchronises the
        current stream.  The counters live in the slot's workspace at a place that depends on the
        batch size: read them between forwards of equally sized batches."""
        if self._fwd_batch is None:
            raise _hip.GvaeHipError("stream_stats called before a forward")
        ptr, cnt = c_void_p(), c_int64()
        _hip._check(self.L.gvae_step_read_debug(ctypes.byref(self.dims), self._fwd_batch[0].num_graphs,
                                                self._ws.data_ptr(), 17, ctypes.byref(ptr), ctypes.byref(cnt)))
        return _hip.mpm_stream_stats(ptr.value, reset)

    # ------------------------------------------------------------------ inspection (tests)
    _DEBUG = {"vec_logits": 0, "node_logits": 1, "P": 2, "X": 3, "mu": 4, "logvar_raw": 5, "rec_i": 6,
              "kl_i": 7, "status": 8, "Y1": 9, "H1": 10, "Y2": 11, "H2": 12, "gvec": 13, "Ahat": 14,
              "dY1": 15, "dgvec": 16}

    def debug_tensor(self, name: str, B: int) -> torch.Tensor:
        """Copy of an intermediate buffer of the last forward of B graphs."""
        ptr, cnt = c_void_p(), c_int64()
        _hip._check(self.L.gvae_step_read_debug(ctypes.byref(self.dims), B, self._ws.data_ptr(),
                                                self._DEBUG[name], ctypes.byref(ptr), ctypes.byref(cnt)))
        off = ptr.value - self._ws.data_ptr()
        raw = self._ws[off: off + cnt.value * 4]
        if name == "status":
            return raw.view(torch.int32).clone()
        t = raw.view(torch.float32).clone()
        N, Z = self.N, self.Z
        shapes = {"vec_logits": (B, N * (N - 1) // 2), "node_logits": (B, N), "P": (B, N, N),
                  "X": (B, N, N), "mu": (B, Z), "logvar_raw": (B, Z), "rec_i": (B,), "kl_i": (B,),
                  "Y1": (B, N, self.H), "H1": (B, N, self.H), "Y2": (B, N, self.H), "H2": (B, N, self.H),
                  "gvec": (B, self.H), "Ahat": (B, N, N), "dY1": (B, N, self.H), "dgvec": (B, self.H)}
        return t.view(shapes[name])
