/*
 * C ABI of libgvae_hip.so: the MI355X (gfx950) kernels behind the GraphVAE
 * training step.  Plain pointers and sizes only; every pointer named `d_*` or
 * documented as "device" is a HIP device pointer owned by the caller.
 *
 * Conventions
 *   - every function returns 0 on success, a negative GVAE_E* code otherwise;
 *     gvae_last_error() returns the text of the calling thread's last failure.
 *   - all launches go to `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream).  Functions are asynchronous and never synchronise unless
 *     they say so.  The library allocates no device memory: scratch comes in
 *     through explicit workspace arguments.
 *   - batched graph matrices are row-major (n_graphs, ld, ld) fp32 blocks; graph g
 *     starts at `base + g*stride` (elements) and only its leading R[g] x R[g]
 *     block is read or written.  R is an int32 device array.
 *   - "canonical order": the fp32 summation orders of the reference's CPU path
 *     (documented in csrc/gvae_canonical.h); results are bit-identical to it.
 *
 * Each entry point cites the reference code it replaces (paths relative to the
 * reference repository root).
 */
#ifndef GVAE_HIP_H
#define GVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVAE_OK 0
#define GVAE_EINVAL (-1)   /* bad argument (size, alignment, null pointer)        */
#define GVAE_ELAUNCH (-2)  /* HIP reported a launch/runtime error                 */
#define GVAE_ELIMIT (-3)   /* size beyond what the kernels support                */

/* Largest node count the LDS-resident kernels (fused MPM, assignment) accept. */
#define GVAE_MAX_NODES_LDS 128

int gvae_version(void);
const char *gvae_last_error(void);

/* Record HIP events around every launch of the named kernel families so that a
 * caller can read accumulated device time per family.  Off by default.
 * gvae_profile_read synchronises the recorded events (host blocks).            */
int gvae_profile_enable(int on);
int gvae_profile_reset(void);
/* family: 0 = edge_sim, 1 = mpm_stream, 2 = unused, 3 = mpm_fused, 4 = lsap,
 *         5 = node_sim                                                          */
int gvae_profile_read(int family, double *total_ms, int64_t *launches);

/* ---------------------------------------------------------------------------
 * Node-similarity block.  Replaces graphvae/models/model.py:184-195 (+54-56):
 * unit diagonals, degrees (degB = soft degree, canonical order) and
 * NS[i,a] = 1/(|degA[i]-degB[a]|+1).  d_NS has the same (stride, ld) layout as A.
 */
int gvae_node_sim(const float *d_A, const float *d_P, const int32_t *d_R, int n_graphs,
                  int ld, int64_t stride, float *d_NS, void *stream);

/* Layouts of a graph's block in the S buffer.  DENSE is the reference's (R,R,R,R)
 * tensor.  PADDED stores every row S[i,j,a,:] in RS = 4*ceil(R/4) floats (the tail
 * repeats the row's last element, which cannot change a row maximum), so that all
 * rows are 16-byte aligned whatever R is; the block then holds R*R*R*RS floats. */
#define GVAE_S_DENSE 0
#define GVAE_S_PADDED 1

/* ---------------------------------------------------------------------------
 * Edge-similarity tensor, materialised.  Replaces GraphVAE._edge_sim_tensor,
 * graphvae/models/model.py:171-216.  Graph g's tensor is written contiguously as
 * (R,R,R,R) fp32 (rows of RS floats for GVAE_S_PADDED), index order (i,j,a,b), at
 * d_S + d_S_off[g] (element offsets; each must be a multiple of 4).  d_NS comes
 * from gvae_node_sim.  r_max >= max R.
 */
int gvae_edge_sim(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                  int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                  const int64_t *d_S_off, void *stream);

/* ---------------------------------------------------------------------------
 * Max-pool matching on a materialised S (streams S from HBM once per iteration).
 * Replaces GraphVAE._mpm, graphvae/models/model.py:247-276.  d_X holds X0 on
 * entry and the result on exit (layout (n_graphs, ldx, ldx), stride strideX).
 * The whole batch runs in one launch: the per-iteration host sync of the reference
 * (model.py:273) is replaced by a device-side work queue in which every graph
 * advances on its own and leaves as soon as it has converged.
 *   s_layout      GVAE_S_DENSE or GVAE_S_PADDED, as the blocks at d_S_off were written
 *   d_iters_done  int32[n_graphs] out: iterations executed per graph; negative for every
 *                 graph if the work queue faulted (a bounded wait ran out: X is invalid)
 *   d_ws          workspace, gvae_mpm_stream_ws_bytes(n_graphs, ldx) bytes
 */
int64_t gvae_mpm_stream_ws_bytes(int n_graphs, int ldx);
int gvae_mpm_stream(const float *d_S, const int64_t *d_S_off, int s_layout, const int32_t *d_R,
                    int n_graphs, int r_max, float *d_X, int ldx, int64_t strideX, int iters,
                    float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream);

/* One MPM iteration before its normalisation, Y[i,a] = X[i,a]*S[i,i,a,a] + sum_{j != i} max_b
 * S[i,j,a,b]*X[j,b], and the gradients of it w.r.t. X (d_gX, overwritten) and, when d_gS is not
 * NULL, w.r.t. S (added onto d_gS): the pieces of the autograd of the public `_mpm`
 * (graphvae/models/model.py:247-276 is differentiable; model_legacy2.py:670-684 checks the
 * gradient w.r.t. X0).  A maximum's gradient goes to the maximising b, in equal shares between
 * ties, as torch's amax does.  One graph, dense (R,R,R,R) S of any values, R <= 255. */
int gvae_mpm_pool_fwd(const float *d_S, int R, const float *d_X, float *d_Y, void *stream);
int gvae_mpm_pool_bwd(const float *d_S, int R, const float *d_X, const float *d_gY, float *d_gX,
                      float *d_gS, void *stream);

/* gvae_mpm_stream without the workspace argument, for callers that cannot allocate device
 * memory (tight serving loops, harnesses that forbid hipMalloc).
 * No counterpart in the reference.  The workspace is a built-in device array of
 * GVAE_STATIC_SCRATCH_BYTES bytes per GPU (held from the first call on that GPU until the
 * process ends); the call allocates and frees nothing and records no profiler events.
 * Accepted sizes: gvae_mpm_stream_ws_bytes(n_graphs, ldx) <= GVAE_STATIC_SCRATCH_BYTES (e.g. 340
 * graphs at ldx 32, 100 at ldx 64, 28 at ldx 128); larger ones fail with GVAE_ELIMIT.  Every call
 * on a GPU shares that GPU's array: calls on one GPU must not overlap (two streams, two
 * threads), they would race on the queue.  Same X and iteration counts as gvae_mpm_stream.
 * The first call on a GPU looks the array's address up (and, for r_max above 64 or so, raises the
 * kernel's dynamic-LDS limit): host-side set-up that must not happen inside a stream capture, and
 * the call has not been exercised under capture at all.  The first 16 bytes of the array are the
 * kernel's statistics words (gvae_mpm_stream_stats); they start at zero with the process and
 * only ever accumulate. */
#define GVAE_STATIC_SCRATCH_BYTES (2 << 20)
int gvae_mpm_stream_static(const float *d_S, const int64_t *d_S_off, int s_layout, const int32_t *d_R,
                           int n_graphs, int r_max, float *d_X, int ldx, int64_t strideX, int iters,
                           float tol, int early_stop, int32_t *d_iters_done, void *stream);

/* gvae_edge_sim followed by gvae_mpm_stream from the uniform start X0 = 1/R of
 * graphvae/models/model.py:317, in one call (what the step's streamed matching uses).  The
 * tensor is written as by gvae_edge_sim; the FIRST iteration is computed without reading it
 * back: with a uniform X0, max_b S[i,j,a,b]*x0 = (max_b S[i,j,a,b])*x0 exactly (rounding is
 * monotone), and the row maximum of S is max(0, A[i,j]*max_{b!=a} P[a,b]) (the minimum over b
 * when A[i,j] < 0), so that iteration costs O(R^3) per graph instead of one pass over S.  The
 * SECOND iteration of a graph of at most 64 or at least 89 nodes is taken by the kernel that writes the tensor,
 * from the values on their way to memory (the work-group that produces S[i,:,:,:] computes row
 * i of the update), which saves the first pass over S.  Every later iteration streams S, except
 * those of the latency-bound tail, which the S-free kernel takes over
 * (gvae_set_stream_handoff_bytes).  d_X is output only.  Same X, same iteration counts as the
 * two calls, bit for bit.  d_ws: gvae_mpm_stream_ws_bytes. */
int gvae_match_stream(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                      int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                      const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                      float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream);

/* Tail hand-off of gvae_match_stream.  With early stop, the last unconverged graphs of a batch
 * iterate on a nearly empty device, where an iteration costs the latency of the queue kernel's
 * chain of dependent steps (about 19 us), not HBM time.  Once the similarity tensors of the
 * graphs still iterating add up to less than `bytes` (default 256 MiB: two chain latencies at
 * the streaming rate) plus four times the graph's own tensor (alone on the device a graph
 * streams at about a quarter of the device's rate), a graph of at most 88 nodes (an S-free
 * iteration takes 5 us at 32 nodes, 34 us at 88) with a 0/1 adjacency leaves the queue at
 * the end of its current iteration and the S-free kernel runs its remaining iterations after
 * the streaming launch: same X and iteration counts, bit for bit.  0 = never hand off.  Applies
 * to gvae_match_stream and the streamed step only (gvae_mpm_stream takes an arbitrary S, for
 * which no S-free form exists).  Process-wide setting; no counterpart in the reference. */
int gvae_set_stream_handoff_bytes(int64_t bytes);

/* What the streaming kernel has executed through this workspace (a gvae_mpm_stream_ws_bytes
 * buffer; the counters sit in its first 16 bytes) since the last reset: the algorithmic bytes
 * sum(4 R^4 + 12 R^2) and the number of graph-iterations it streamed (iterations done by the
 * first-iteration shortcut or by the S-free tail are not among them).  Copies the two counters
 * to the host, synchronising `stream`; reset != 0 then clears them (a fresh workspace holds
 * garbage: reset it once before measuring).  Measurement aid. */
int gvae_mpm_stream_stats(void *d_ws, int64_t *h_bytes, int64_t *h_graph_iterations, int reset,
                          void *stream);

/* ---------------------------------------------------------------------------
 * Max-pool matching without S: same recurrence and same bits as
 * gvae_edge_sim + gvae_mpm_stream when A is 0/1 with unit diagonal and P >= 0,
 * using  max_b S[i,j,a,b] X[j,b] = A[i,j] * max_{b!=a} P[a,b] X[j,b]  (j != i).
 * One work-group per graph, everything resident in LDS.  r_max <= GVAE_MAX_NODES_LDS
 * (and the LDS budget: 12*r_max^2 + 16*r_max + 4*r_max*ceil(r_max/32) bytes <= 159 KiB).
 * No counterpart in the reference (it always materialises S).
 */
int gvae_mpm_fused(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                   int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                   int64_t strideX, int iters, float tol, int early_stop,
                   int32_t *d_iters_done, void *stream);

/* ---------------------------------------------------------------------------
 * Assignment.  Replaces the host round trip of graphvae/models/model.py:319-323:
 * cost = -X in double, scipy.optimize.linear_sum_assignment (SciPy 1.15.3
 * shortest-augmenting-path, identical scan and tie-breaking order), then
 * perm[col[i]] = i, extended with the identity on [R, n_perm).
 *   d_perm   int64 (n_graphs, n_perm) out
 *   d_status int32[n_graphs] out: 0 ok, 1 = non-finite/infeasible cost
 */
int gvae_lsap(const float *d_X, int ldx, int64_t strideX, const int32_t *d_R, int n_graphs,
              int r_max, int64_t *d_perm, int n_perm, int32_t *d_status, void *stream);

/* gvae_mpm_fused followed by gvae_lsap (graphvae/models/model.py:317-323: the MPM and the
 * assignment on its result).  While every graph's work-group fits on the device at once (1024 /
 * 512 / 256 graphs for r_max <= 32 / 64 / 128) this is ONE launch: the work-group that iterated
 * a graph solves that graph's assignment from the X it holds in LDS, so a graph that converges
 * early is assigned early instead of waiting for the slowest graph of the batch; larger batches
 * run as the two launches.  Same X, same perm, same status either way. */
int gvae_mpm_fused_assign(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                          int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                          int64_t strideX, int iters, float tol, int early_stop,
                          int32_t *d_iters_done, int64_t *d_perm, int n_perm, int32_t *d_status,
                          void *stream);


/* ===========================================================================
 * Whole training step (dense encoder/decoder + matching + losses + optimiser).
 * =========================================================================== */

/* Model dimensions: in_dim, hid_dim, z_dim, max_nodes of
 * GraphVAE.__init__, graphvae/models/model.py:63-94. */
typedef struct gvae_dims {
  int32_t F, H, Z, N;
} gvae_dims;

/* Parameter tensors, in the order of the reference module's named_parameters()
 * (graphvae/models/model.py:71-89; GCNConv registers `bias` before `lin.weight`).
 * All live in ONE flat fp32 buffer; each tensor starts at a multiple of 4 floats. */
enum {
  GVAE_P_CONV1_B, GVAE_P_CONV1_W, GVAE_P_BN1_W, GVAE_P_BN1_B,
  GVAE_P_CONV2_B, GVAE_P_CONV2_W, GVAE_P_BN2_W, GVAE_P_BN2_B,
  GVAE_P_MU_W, GVAE_P_MU_B, GVAE_P_LV_W, GVAE_P_LV_B,
  GVAE_P_DE1_W, GVAE_P_DE1_B, GVAE_P_DE2_W, GVAE_P_DE2_B,
  GVAE_P_DN1_W, GVAE_P_DN1_B, GVAE_P_DN2_W, GVAE_P_DN2_B,
  GVAE_P_COUNT
};

/* Fills offsets[GVAE_P_COUNT] / sizes[GVAE_P_COUNT] (floats; either may be NULL) and
 * returns the length of the flat buffer in floats (padding included), -1 on bad dims. */
int64_t gvae_param_layout(const gvae_dims *d, int64_t *offsets, int64_t *sizes);

#define GVAE_MATCH_NONE 0    /* identity permutation, no matching kernels           */
#define GVAE_MATCH_STREAM 1  /* materialise S, stream it per MPM iteration (HBM)    */
#define GVAE_MATCH_FUSED 2   /* S-free MPM resident in LDS                          */

#define GVAE_LOSS_CURRENT 0       /* GraphVAE.forward, graphvae/models/model.py:337-343       */
#define GVAE_LOSS_DUMMY_MASKED 1  /* GraphVAE.forward_old, model.py:403-408,425-427: the node
                                     logits of padded entries are overwritten with -10 before
                                     the node loss (constant term, no gradient)               */
#define GVAE_LOSS_INDIVIDUAL_RHO 2 /* GraphVAE.forward_individual_rho, graphvae/models/model_legacy.py:257-363, on
                                    * the current heads (its decoder `dec` is `dec_edge`; `dec_node` takes no part):
                                    * matching over all N rows of the padded matrices (r_max must be N), ground
                                    * truth permuted and prediction left in place, positives weighted by each
                                    * graph's own density; the backward reads the permutation, so it has to
                                    * follow the matching of the same batch */

/* Bytes of scratch gvae_step_forward/backward need for n_graphs graphs.  The
 * forward leaves its activations there for the backward of the same batch. */
int64_t gvae_step_ws_bytes(const gvae_dims *d, int n_graphs);

/* Forward of GraphVAE.forward, graphvae/models/model.py:281-373, for a whole batch.
 *   d_params     flat parameters (gvae_param_layout)
 *   d_bn_state   [bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var],
 *                4*H floats; updated in place when train != 0, graph by graph in batch
 *                order as the reference's per-graph BatchNorm calls do
 *   d_x          (n_graphs, N, F) node features, zero rows beyond n_x[g]
 *   d_adj        (n_graphs, N, N) adj_dense: 0/1, symmetric, unit diagonal on real nodes
 *   d_n_x        feature rows per graph (N for `deg` features, R for `id`;
 *                graphvae/train/train_ddp.py:67-78)
 *   d_R          real nodes per graph;  r_max >= max R (host value)
 *   d_eps        (n_graphs, Z) standard-normal draws for the reparameterisation
 *   d_pos_weight 1 float: the pos_weight_global buffer
 *   loss_variant GVAE_LOSS_*; the backward of the batch must be given the same value
 *   matching     GVAE_MATCH_*;  STREAM needs d_S and d_S_off (both may be NULL otherwise).
 *                S is kept in the GVAE_S_PADDED layout and may be smaller than the batch's
 *                whole tensor: the graphs are matched in n_chunks consecutive chunks,
 *                chunk c = graphs [h_chunk_begin[c], h_chunk_begin[c+1]), all reusing d_S;
 *                d_S_off[g] is graph g's offset from the start of d_S within its chunk
 *                (so it restarts at 0 at every chunk).  h_chunk_begin is a HOST array of
 *                n_chunks+1 entries; NULL with n_chunks = 0 means one chunk.
 *   d_out4       [rec, kl, max_logit, logvar_max]   (loss = rec + kl)
 *   d_perm       (n_graphs, N) int64 perm_full per graph;  d_iters_done MPM iterations
 */
int gvae_step_forward(const gvae_dims *d, const float *d_params, float *d_bn_state, const float *d_x,
                      const float *d_adj, const int32_t *d_n_x, const int32_t *d_R, int n_graphs,
                      int r_max, const float *d_eps, const float *d_pos_weight, int loss_variant,
                      int train, int matching, int mpm_iters, float tol, int early_stop, float *d_S,
                      const int64_t *d_S_off, const int32_t *h_chunk_begin, int n_chunks,
                      void *d_ws, float *d_out4, int64_t *d_perm, int32_t *d_iters_done,
                      void *stream);

/* The same forward in two calls, for callers that overlap the matching with the backward.
 * The parameter gradients do not depend on the permutation: in model.py:325-334 prediction and
 * ground truth are gathered through the same perm_full, so the edge loss visits every pair of
 * real nodes exactly once whatever the assignment.  The backward therefore only needs what
 *   gvae_step_forward_model       encoder, heads, reparameterisation, decoder
 *                                 (model.py:295-303)
 * leaves in the workspace, and
 *   gvae_step_forward_match_loss  matching, assignment, losses, diagnostics (model.py:305-373)
 * may be enqueued on a second stream (after an event recorded behind the model part),
 * concurrently with gvae_step_backward of the same batch: it reads the logits, mu and logvar
 * and writes only buffers the backward never touches.  The caller joins the two streams
 * before it reads d_out4 / d_perm and before the next forward reuses the workspace.
 * Arguments as in gvae_step_forward.  `phases` selects what the call enqueues, a run of
 * consecutive parts of
 *   GVAE_PHASE_PREP         probabilities, node block, uniform X0; streamed matching of one chunk:
 *                           also the similarity tensor and the first iteration
 *   GVAE_PHASE_MPM          the remaining MPM iterations (streamed: the persistent launch; a
 *                           chunked batch walks all its chunks through the S buffer here)
 *   GVAE_PHASE_TAIL         streamed matching of one chunk with early stop: the last iterations of
 *                           the slowest graphs, which the persistent launch leaves to the S-free
 *                           kernel (nothing otherwise)
 *   GVAE_PHASE_ASSIGN_LOSS  assignment, losses, diagnostics
 * (GVAE_PHASE_MATCH = the first three).  Calls for one batch must come in this order, each ordered
 * behind the one before (stream order or an event), with the same matching mode, r_max and chunk
 * list.  A caller with a third stream lets tail + assignment of one batch (a work-group or a
 * wavefront per graph, latency-bound) run beside the similarity tensor of the next batch, and
 * orders the next batch's GVAE_PHASE_MPM behind this batch's GVAE_PHASE_TAIL: beside a persistent
 * launch the tail's kernels find no room. */
#define GVAE_PHASE_ASSIGN_LOSS 2
#define GVAE_PHASE_PREP 4
#define GVAE_PHASE_MPM 8
#define GVAE_PHASE_TAIL 16
#define GVAE_PHASE_MATCH (GVAE_PHASE_PREP | GVAE_PHASE_MPM | GVAE_PHASE_TAIL)
int gvae_step_forward_model(const gvae_dims *d, const float *d_params, float *d_bn_state,
                            const float *d_x, const float *d_adj, const int32_t *d_n_x,
                            const int32_t *d_R, int n_graphs, const float *d_eps, int train,
                            void *d_ws, void *stream);
int gvae_step_forward_match_loss(const gvae_dims *d, const float *d_adj, const int32_t *d_R,
                                 int n_graphs, int r_max, const float *d_pos_weight,
                                 int loss_variant, int matching, int mpm_iters, float tol,
                                 int early_stop, float *d_S, const int64_t *d_S_off,
                                 const int32_t *h_chunk_begin, int n_chunks, void *d_ws,
                                 float *d_out4, int64_t *d_perm, int32_t *d_iters_done,
                                 int phases, void *stream);

/* Backward of  c_rec*up[0]*rec + c_kl*up[1]*kl  with respect to every parameter
 * (what loss.backward() does in graphvae/train/train_ddp.py:283-291, where
 * c_rec = 1 and c_kl = beta).  d_up may be NULL (= ones) or 2 device floats.
 * d_ws must be the workspace the forward of this batch just filled.  d_grads is
 * overwritten (flat, same layout as the parameters; padding floats untouched). */
int gvae_step_backward(const gvae_dims *d, const float *d_params, const float *d_x,
                       const float *d_adj, const int32_t *d_n_x, const int32_t *d_R, int n_graphs,
                       const float *d_eps, const float *d_pos_weight, int loss_variant, float c_rec,
                       float c_kl, const float *d_up, void *d_ws, float *d_grads, void *stream);

/* Pointer into the workspace for inspection by tests: which = 0 edge logits (B,U),
 * 1 node logits (B,N), 2 P (B,N,N), 3 X (B,N,N), 4 mu, 5 logvar (pre-clamp) (B,Z),
 * 17 the streamed matching's workspace (for gvae_mpm_stream_stats; count = its bytes / 4),
 * 6 rec_i, 7 kl_i (B), 8 assignment status (B, int32), 9 Y1, 10 H1, 11 Y2, 12 H2 (B,N,H:
 * aggregated pre-BatchNorm values without bias / post-ReLU activations of the two GCN
 * layers), 13 pooled vector (B,H), 14 normalised adjacency (B,N,N), 15 dY of layer 1 after
 * a backward (B,N,H), 16 d(pooled vector) (B,H). */
int gvae_step_read_debug(const gvae_dims *d, int n_graphs, void *d_ws, int which, float **out_ptr,
                         int64_t *out_count);

/* Epoch loss without a host synchronisation per step; replaces `sum_loss_tr += loss.item()`
 * (graphvae/train/train_ddp.py:316-318): d_loss_sum[0] += rec + beta * kl from the four scalars a
 * forward left in d_out4 (float arithmetic as the reference's `loss`, summed in double), and
 * d_out4_copy (nullable, 4 floats) receives a copy of them.  One launch on `stream`. */
int gvae_step_accumulate_loss(const float *d_out4, float beta, float *d_out4_copy, double *d_loss_sum,
                              void *stream);

/* Global-norm clipping + Adam with L2 weight decay on the flat buffers; replaces
 * clip_grad_norm_ + torch.optim.Adam.step, graphvae/train/train_ddp.py:243,292,316.
 * Gradients are first multiplied by grad_scale (1/world_size after a SUM all-reduce).
 * step counts from 1.  clip_max <= 0 disables clipping.  d_norm_out (nullable)
 * receives the pre-clip gradient norm.  d_ws: gvae_adam_ws_bytes() bytes. */
int64_t gvae_adam_ws_bytes(void);
int gvae_adam_step(float *d_params, const float *d_grads, float *d_m, float *d_v, int64_t n,
                   float grad_scale, float clip_max, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, void *d_ws, float *d_norm_out, void *stream);

/* Decoder pass of the sampler; replaces the per-sample loop of experiments/sample.py:50-73:
 * edge logits = dec_edge(z) (n, N(N-1)/2), node logits = dec_node(z) (n, N); a node exists
 * when sigmoid(node logit) > th_node; adjacency bit (r,c) = sigmoid(v) > th_edge with
 * v = the pair's edge logit, or -10 on the diagonal and where an end point does not exist.
 * Outputs: both logit matrices, d_adj (n,N,N) bytes 0/1, d_exists (n,N) bytes 0/1,
 * d_count (n) = existing nodes per sample.  d_ws: gvae_sample_ws_bytes(d, n) bytes. */
int64_t gvae_sample_ws_bytes(const gvae_dims *d, int n);
int gvae_sample_decode(const gvae_dims *d, const float *d_params, const float *d_z, int n,
                       float th_edge, float th_node, float *d_edge_logits, float *d_node_logits,
                       uint8_t *d_adj, uint8_t *d_exists, int32_t *d_count, void *d_ws,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GVAE_HIP_H */
