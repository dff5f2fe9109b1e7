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
/* family: 0 = edge_sim, 1 = mpm_stream iteration, 2 = mpm_stream normalise,
 *         3 = mpm_fused, 4 = lsap, 5 = node_sim                                 */
int gvae_profile_read(int family, double *total_ms, int64_t *launches);

/* ---------------------------------------------------------------------------
 * Node-similarity block.  Replaces graphvae/models/model.py:184-195 (+54-56):
 * unit diagonals, degrees (degB = soft degree, canonical order) and
 * NS[i,a] = 1/(|degA[i]-degB[a]|+1).  d_NS has the same (stride, ld) layout as A.
 */
int gvae_node_sim(const float *d_A, const float *d_P, const int32_t *d_R, int n_graphs,
                  int ld, int64_t stride, float *d_NS, void *stream);

/* ---------------------------------------------------------------------------
 * Edge-similarity tensor, materialised.  Replaces GraphVAE._edge_sim_tensor,
 * graphvae/models/model.py:171-216.  Graph g's tensor is written contiguously as
 * (R,R,R,R) fp32, index order (i,j,a,b), at d_S + d_S_off[g] (element offsets;
 * each must be a multiple of 4).  d_NS comes from gvae_node_sim.  r_max >= max R.
 */
int gvae_edge_sim(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                  int n_graphs, int ld, int64_t stride, int r_max, float *d_S,
                  const int64_t *d_S_off, void *stream);

/* ---------------------------------------------------------------------------
 * Max-pool matching on a materialised S (streams S from HBM once per iteration).
 * Replaces GraphVAE._mpm, graphvae/models/model.py:247-276.  d_X holds X0 on
 * entry and the result on exit (layout (n_graphs, ldx, ldx), stride strideX).
 * The per-iteration host sync of the reference (model.py:273) is replaced by a
 * per-graph device flag: a converged graph's work-groups exit at once.
 *   d_iters_done  int32[n_graphs] out: iterations executed per graph
 *   d_ws          workspace, gvae_mpm_stream_ws_bytes(n_graphs, ldx) bytes
 */
int64_t gvae_mpm_stream_ws_bytes(int n_graphs, int ldx);
int gvae_mpm_stream(const float *d_S, const int64_t *d_S_off, const int32_t *d_R, int n_graphs,
                    int r_max, float *d_X, int ldx, int64_t strideX, int iters, float tol,
                    int early_stop, int32_t *d_iters_done, void *d_ws, void *stream);

/* ---------------------------------------------------------------------------
 * Max-pool matching without S: same recurrence and same bits as
 * gvae_edge_sim + gvae_mpm_stream when A is 0/1 with unit diagonal and P >= 0,
 * using  max_b S[i,j,a,b] X[j,b] = A[i,j] * max_{b!=a} P[a,b] X[j,b]  (j != i).
 * One work-group per graph, everything resident in LDS.  r_max <= GVAE_MAX_NODES_LDS
 * (and the LDS budget: 12*r_max^2 + 4*r_max*ceil(r_max/32) bytes <= 160 KiB).
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

#ifdef __cplusplus
}
#endif
#endif /* GVAE_HIP_H */
