// Internal launch helpers of the dense (encoder / decoder / loss / optimiser)
// kernels, shared between dense.hip and step.hip.
#pragma once
#include "gvae_common.h"

namespace gvae {

// C[b] (M x N, row stride c_rs, unit column stride)
//   = op(A[b]) (M x K, element (m,k) at A + m*a_rs + k*a_cs)
//   * op(B[b]) (K x N, element (k,n) at B + k*b_rs + n*b_cs)   [+ bias[n]] [relu] [+= C]
// fp32 in, fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32), which is a
// k-ordered fp32 fma chain, i.e. plain fp32 numerics.
struct GemmDesc {
  const float *A = nullptr, *B = nullptr, *bias = nullptr;
  float *C = nullptr;
  int M = 0, N = 0, K = 0;
  int64_t a_rs = 0, a_cs = 0, b_rs = 0, b_cs = 0, c_rs = 0;
  int batch = 1;
  int64_t a_bs = 0, b_bs = 0, c_bs = 0;
  int relu = 0, accumulate = 0;
  // optional (M x N, same strides as C): the result is zeroed where act <= 0 (backward of a ReLU
  // whose output is `act`), after bias / accumulation
  const float *relu_mask_act = nullptr;
  // split-K (batch == 1 only): partial products go to `splitk_ws`
  // (splits * M * N floats) and are then summed in split order.
  int splits = 1;
  float *splitk_ws = nullptr;
};
int launch_gemm(const GemmDesc &d, hipStream_t s);

// Up to GEMM_GROUP_MAX GEMMs that do not depend on each other, in one launch (plus one
// split-K reduction launch if any member is split).  launch_gemm_group hands each split
// member its own part of the workspace (`splitk_floats` floats at `splitk_base`).
#define GEMM_GROUP_MAX 6
struct GemmGroup {
  GemmDesc d[GEMM_GROUP_MAX];
  int n = 0;
  // the launch is one-dimensional: member k owns the work-groups [wg_begin[k], wg_begin[k+1]),
  // numbered z-major over its tiles_x * tiles_y * (splits or batch) tiles; likewise the
  // split-K reduction owns [red_begin[k], red_begin[k+1]) (empty for an unsplit member)
  int wg_begin[GEMM_GROUP_MAX + 1];
  int red_begin[GEMM_GROUP_MAX + 1];
  int tiles_x[GEMM_GROUP_MAX], tiles_y[GEMM_GROUP_MAX];
};
int launch_gemm_group(GemmDesc *descs, int n, float *splitk_base, int64_t splitk_floats,
                      hipStream_t s);

// The model half of the forward (encoder, pooling, heads, reparameterisation, first decoder
// layers) as one launch with a work-group per graph: see model_fwd_graph_kernel.
struct ModelFwdArgs {
  const float *x, *adj, *eps;
  const int32_t *n_x, *R;
  const float *W1, *b1, *g1, *be1, *rm1, *rv1, *W2, *b2, *g2, *be2, *rm2, *rv2;
  const float *Wmu, *bmu, *Wlv, *blv, *Wde1, *bde1, *Wdn1, *bdn1;
  float *Ahat, *H0a, *Y1, *H1, *H0b, *Y2, *H2;
  float *mean1, *invstd1, *varu1, *mean2, *invstd2, *varu2;
  float *gvec, *mu, *lvraw, *z, *he, *hn;
  double *T64, *Y64;   // (B,N,H) each: fp64 activations of the any-size kernel
  int32_t *bn_mode;
  int N, F, H, Z, train;
};
int launch_model_fwd_graph(const ModelFwdArgs &a, int B, hipStream_t s);

// Sizes the LDS-resident kernels of the model chain (forward and backward) take.
bool model_chain_resident(int N, int H, int Z);
// Backward of the model chain for one graph per work-group (model_bwd_lds_kernel): from the
// gradients of the decoder's hidden layers (already masked by their ReLUs) to the gradients of
// the two GCN layers' linear outputs, with the BatchNorm parameters' per-graph partial sums.
struct ModelBwdArgs {
  const float *dhe, *dhn;
  const float *Wde1, *Wdn1, *Wmu, *Wlv, *W2;
  const float *mu, *lvraw, *eps, *d_up;
  const float *Ahat, *Y1, *H1, *Y2, *H2;
  const float *b1, *g1, *mean1, *invstd1, *b2, *g2, *mean2, *invstd2;
  const int32_t *n_x, *R, *bn_mode;
  float *dz, *dmu, *dlv, *dgvec, *dY, *dH0_1, *dH0_2;
  float *dgam1, *dbet1, *dbias1, *dgam2, *dbet2, *dbias2;
  float ck;   // KL coefficient / number of graphs
  int N, H, Z;
};
int launch_model_bwd_graph(const ModelBwdArgs &a, int B, hipStream_t s);

// running statistics of up to two BatchNorm layers after a training forward, in one launch
struct BnRunningArgs {
  const float *mean[2], *var_unb[2];
  float *run_mean[2], *run_var[2];
  int layers, B, H;
  float momentum;
};
int launch_bn_running_update(const BnRunningArgs &a, hipStream_t s);
// Backward of BatchNorm (over each graph's n_x live rows) + ReLU: dY (B,N,H) from dOut; per-graph
// partial sums dgamma_p, dbeta_p, dbias_p (B,H).  Y holds the CENTRED pre-activation the forward
// left there.  mode (device): non-zero if the forward used batch statistics, zero if it used the
// running ones.
int launch_bn_relu_bwd(const float *dOut, const float *Y, const float *bias, const float *out,
                       const float *gamma, const float *mean, const float *invstd,
                       const int32_t *n_x, int B, int N, int H, const int32_t *mode, float *dY,
                       float *dgamma_p, float *dbeta_p, float *dbias_p, hipStream_t s);
// out[c] (+)= sum_r M[r*ld + c]   (rows summed in index order)
int launch_colsum(const float *M, int rows, int cols, int64_t ld, float *out, int accumulate,
                  hipStream_t s);
// out[k][c] = sum_r M[k][r*ld[k] + c] for up to COLSUM_GROUP_MAX matrices in one launch
#define COLSUM_GROUP_MAX 12
struct ColsumGroup {
  const float *M[COLSUM_GROUP_MAX];
  float *out[COLSUM_GROUP_MAX];
  int rows[COLSUM_GROUP_MAX], cols[COLSUM_GROUP_MAX];
  int64_t ld[COLSUM_GROUP_MAX];
  int n = 0;
  void add(const float *m, int r, int c, int64_t l, float *o) {
    M[n] = m; rows[n] = r; cols[n] = c; ld[n] = l; out[n] = o; ++n;
  }
};
int launch_colsum_group(const ColsumGroup &g, hipStream_t s);
int launch_pool_bwd(const float *dgvec, const int32_t *n_x, const int32_t *R, int B, int N, int H,
                    float *dH, hipStream_t s);
// dmu, dlvraw from dz and the KL term (coefficient c_kl / B per graph)
int launch_reparam_bwd(const float *dz, const float *mu, const float *lvraw, const float *eps, int B,
                       int Z, float c_kl, const float *d_up, float *dmu, float *dlvraw,
                       hipStream_t s);
int launch_prob(const float *vec, const float *node, int B, int N, float *P, hipStream_t s);
int launch_loss_fwd(const float *vec, const float *node, const float *adj, const float *mu,
                    const float *lvraw, const int64_t *perm, const int32_t *R, int B, int N, int Z,
                    const float *pos_weight, int mask_dummy, float *rec_i, float *kl_i,
                    float *maxlogit_i, float *out4, hipStream_t s);
int launch_loss_bwd(const float *vec, const float *node, const float *adj, const int32_t *R, int B,
                    int N, const float *pos_weight, int mask_dummy, float c_rec, const float *d_up,
                    float *dvec, float *dnode, hipStream_t s);
int launch_loss_fwd_indiv(const float *vec, const float *adj, const float *mu, const float *lvraw,
                          const int64_t *perm, const int32_t *R, int B, int N, int Z, float *rec_i,
                          float *kl_i, float *maxlogit_i, float *pw_i, float *out4, hipStream_t s);
int launch_loss_bwd_indiv(const float *vec, const float *adj, const int32_t *R, const int64_t *perm,
                          const float *pw_i, int B, int N, float c_rec, const float *d_up, float *dvec,
                          float *dnode, hipStream_t s);

}  // namespace gvae
