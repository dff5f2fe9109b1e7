// Orchestration of one GraphVAE training step on a batch of padded dense graphs:
// forward (encoder -> heads -> decoder -> matching -> assignment -> losses),
// backward (all parameter gradients into one flat buffer) and the fused
// clip + Adam update.  Every buffer comes from the caller; nothing is allocated.
//
// Reference: graphvae/models/model.py:281-373 (forward), graphvae/train/train_ddp.py:282-316
// (loss weighting, clipping, Adam).
#include "gvae_dense.h"

namespace gvae {

static inline int64_t align4(int64_t n) { return (n + 3) & ~(int64_t)3; }

static int64_t param_layout(const gvae_dims *d, int64_t *off, int64_t *size) {
  const int64_t F = d->F, H = d->H, Z = d->Z, N = d->N, U = N * (N - 1) / 2;
  const int64_t sz[GVAE_P_COUNT] = {
      H, H * F, H, H,          // conv1.bias, conv1.lin.weight, bn1.weight, bn1.bias
      H, H * H, H, H,          // conv2.bias, conv2.lin.weight, bn2.weight, bn2.bias
      Z * H, Z, Z * H, Z,      // mu.weight, mu.bias, logvar.weight, logvar.bias
      H * Z, H, U * H, U,      // dec_edge.0.weight/.bias, dec_edge.2.weight/.bias
      H * Z, H, N * H, N};     // dec_node.0.weight/.bias, dec_node.2.weight/.bias
  int64_t tot = 0;
  for (int i = 0; i < GVAE_P_COUNT; ++i) {
    if (off) off[i] = tot;
    if (size) size[i] = sz[i];
    tot += align4(sz[i]);
  }
  return tot;
}

struct Workspace {
  float *Ahat, *H0a, *Y1, *H1, *H0b, *Y2, *H2;
  float *mean1, *invstd1, *varu1, *mean2, *invstd2, *varu2;
  float *gvec, *mu, *lvraw, *z, *he, *hn, *vec, *node;
  float *P, *NS, *X;
  float *rec_i, *kl_i, *maxlogit_i;
  // backward
  float *dvec, *dnode, *dhe, *dhn, *dz, *dmu, *dlv, *dgvec, *dAct, *dY, *dH0;
  float *dgam_p, *dbet_p, *dbias_p;
  // the per-graph backward keeps both layers' results until the batch-wide sums that follow it
  float *dH0_2, *dgam_p2, *dbet_p2, *dbias_p2;
  float *splitk;
  int64_t splitk_floats;
  int32_t *status;
  int32_t *bn_mode;   // [0]: the forward of this batch normalised with batch statistics
  // GVAE_LOSS_INDIVIDUAL_RHO: node count N for every graph (the matching runs over all rows),
  // the per-graph positive weight and the permutation as the forward left them for the backward
  int32_t *r_full;
  float *pw_i;
  int64_t *perm_fwd;
  char *mpm_ws;
  int64_t bytes;
};

static const int kMaxSplits = 64;

static Workspace carve(const gvae_dims *d, int B, char *base) {
  const int64_t F = d->F, H = d->H, Z = d->Z, N = d->N, U = N * (N - 1) / 2;
  (void)F;
  Workspace w{};
  int64_t off = 0;
  auto take = [&](int64_t nfloat) {
    float *p = reinterpret_cast<float *>(base + off);
    off += (nfloat * 4 + 255) & ~(int64_t)255;
    return p;
  };
  const int64_t BNH = (int64_t)B * N * H, BNN = (int64_t)B * N * N, BH = (int64_t)B * H;
  w.Ahat = take(BNN); w.H0a = take(BNH); w.Y1 = take(BNH); w.H1 = take(BNH);
  w.H0b = take(BNH); w.Y2 = take(BNH); w.H2 = take(BNH);
  w.mean1 = take(BH); w.invstd1 = take(BH); w.varu1 = take(BH);
  w.mean2 = take(BH); w.invstd2 = take(BH); w.varu2 = take(BH);
  w.gvec = take(BH); w.mu = take((int64_t)B * Z); w.lvraw = take((int64_t)B * Z);
  w.z = take((int64_t)B * Z); w.he = take(BH); w.hn = take(BH);
  w.vec = take((int64_t)B * U); w.node = take((int64_t)B * N);
  w.P = take(BNN); w.NS = take(BNN); w.X = take(BNN);
  w.rec_i = take(B); w.kl_i = take(B); w.maxlogit_i = take(B);
  w.dvec = take((int64_t)B * U); w.dnode = take((int64_t)B * N);
  w.dhe = take(BH); w.dhn = take(BH); w.dz = take((int64_t)B * Z);
  w.dmu = take((int64_t)B * Z); w.dlv = take((int64_t)B * Z); w.dgvec = take(BH);
  w.dAct = take(BNH); w.dY = take(BNH); w.dH0 = take(BNH);
  w.dgam_p = take(BH); w.dbet_p = take(BH); w.dbias_p = take(BH);
  w.dH0_2 = take(BNH); w.dgam_p2 = take(BH); w.dbet_p2 = take(BH); w.dbias_p2 = take(BH);
  // split-K partials: splits <= ceil(512 / tiles) and tiles >= M*N/4096, so
  // splits*M*N <= 512*4096 + M*N for every GEMM that is split
  int64_t mn = U * H;
  const int64_t cand[] = {(int64_t)B * H, H * H, H * F, N * H, Z * H, (int64_t)B * Z};
  for (int64_t c : cand) mn = c > mn ? c : mn;
  w.splitk_floats = GEMM_GROUP_MAX * (512 * 4096 + mn);   // room for a whole group of split GEMMs
  w.splitk = take(w.splitk_floats);
  w.status = reinterpret_cast<int32_t *>(take(B));
  w.bn_mode = reinterpret_cast<int32_t *>(take(4));
  w.r_full = reinterpret_cast<int32_t *>(take(B));
  w.pw_i = take(B);
  w.perm_fwd = reinterpret_cast<int64_t *>(take(2 * (int64_t)B * N));
  w.mpm_ws = reinterpret_cast<char *>(take((gvae_mpm_stream_ws_bytes(B, (int)N) + 3) / 4));
  w.bytes = off;
  return w;
}

struct Params {
  const float *p[GVAE_P_COUNT];
};
static Params bind(const gvae_dims *d, const float *flat) {
  int64_t off[GVAE_P_COUNT];
  param_layout(d, off, nullptr);
  Params P;
  for (int i = 0; i < GVAE_P_COUNT; ++i) P.p[i] = flat + off[i];
  return P;
}

// Split-K factor: enough work-groups to fill the chip (256 CUs), at least two 64-deep K-tiles
// per split.
static int pick_splits(int M, int N, int K) {
  const int tiles = ceil_div(M, 64) * ceil_div(N, 64);
  if (tiles >= 256 || K < 256) return 1;
  int s = K / 128;
  const int want = ceil_div(512, tiles);
  if (s > want) s = want;
  if (s > kMaxSplits) s = kMaxSplits;
  return s < 1 ? 1 : s;
}

// C (M x N) = A (M x K, row-major lda) * W^T, W (N x K) row-major  [+bias][relu]
static GemmDesc desc_linear_fwd(const float *A, int64_t lda, const float *W, const float *bias,
                                float *C, int M, int N, int K, int relu) {
  GemmDesc g;
  g.A = A; g.a_rs = lda; g.a_cs = 1;
  g.B = W; g.b_rs = 1; g.b_cs = K;
  g.C = C; g.c_rs = N; g.bias = bias; g.M = M; g.N = N; g.K = K; g.relu = relu;
  return g;
}
// dA (M x K) (+)= dC (M x N) * W, W (N x K) row-major
static GemmDesc desc_linear_bwd_input(const float *dC, const float *W, float *dA, int M, int N, int K,
                                      int accumulate, float *splitk) {
  GemmDesc g;
  g.A = dC; g.a_rs = N; g.a_cs = 1;
  g.B = W; g.b_rs = K; g.b_cs = 1;
  g.C = dA; g.c_rs = K; g.M = M; g.N = K; g.K = N; g.accumulate = accumulate;
  g.splits = pick_splits(M, K, N);
  g.splitk_ws = splitk;
  return g;
}
// dW (N x K) = dC^T (N x M) * A (M x K)
static GemmDesc desc_linear_bwd_weight(const float *dC, const float *A, int64_t lda, float *dW, int M,
                                       int N, int K, float *splitk) {
  GemmDesc g;
  g.A = dC; g.a_rs = 1; g.a_cs = N;
  g.B = A; g.b_rs = lda; g.b_cs = 1;
  g.C = dW; g.c_rs = K; g.M = N; g.N = K; g.K = M;
  g.splits = pick_splits(N, K, M);
  g.splitk_ws = splitk;
  return g;
}
static int linear_fwd(const float *A, int64_t lda, const float *W, const float *bias, float *C, int M,
                      int N, int K, int relu, hipStream_t s) {
  return launch_gemm(desc_linear_fwd(A, lda, W, bias, C, M, N, K, relu), s);
}
static int linear_bwd_input(const float *dC, const float *W, float *dA, int M, int N, int K,
                            int accumulate, float *splitk, hipStream_t s) {
  return launch_gemm(desc_linear_bwd_input(dC, W, dA, M, N, K, accumulate, splitk), s);
}
static int linear_bwd_weight(const float *dC, const float *A, int64_t lda, float *dW, int M, int N,
                             int K, float *splitk, hipStream_t s) {
  return launch_gemm(desc_linear_bwd_weight(dC, A, lda, dW, M, N, K, splitk), s);
}
// Y[b] (N x H) = Ahat[b] (N x N) * X[b] (N x H)
static int aggregate(const float *Ahat, const float *X, float *Y, int B, int N, int H, hipStream_t s) {
  GemmDesc g;
  g.A = Ahat; g.a_rs = N; g.a_cs = 1; g.a_bs = (int64_t)N * N;
  g.B = X; g.b_rs = H; g.b_cs = 1; g.b_bs = (int64_t)N * H;
  g.C = Y; g.c_rs = H; g.c_bs = (int64_t)N * H;
  g.M = N; g.N = H; g.K = N; g.batch = B;
  return launch_gemm(g, s);
}

__global__ void x0_init_kernel(const int32_t *__restrict__ Rarr, int N, float *__restrict__ X) {
  const int g = blockIdx.y;
  const int R = Rarr[g];
  const float v = R > 0 ? (float)(1.0 / (double)R) : 0.0f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += gridDim.x * blockDim.x) {
    const int i = e / N, a = e - i * N;
    X[(int64_t)g * N * N + e] = (i < R && a < R) ? v : 0.0f;
  }
}

__global__ void fill_i32_kernel(int32_t *a, int n, int32_t v) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) a[e] = v;
}

__global__ void identity_perm_kernel(int64_t *perm, int B, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < B * N) perm[e] = e % N;
}

// running sum of rec + beta * kl over the steps of an epoch (reference train_ddp.py:316-318 adds
// loss.item() on the host, one synchronisation per step), and a copy of the step's four scalars
__global__ void accumulate_loss_kernel(const float *__restrict__ out4, float beta,
                                       float *__restrict__ copy, double *__restrict__ loss_sum) {
  const int t = threadIdx.x;
  if (copy && t < 4) copy[t] = out4[t];
  if (t == 0) loss_sum[0] += (double)(out4[0] + beta * out4[1]);
}

// ---- optimiser -------------------------------------------------------------------
#define NORM_BLOCKS 256
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float *__restrict__ g, int64_t n,
                                                          float scale, double *__restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)NORM_BLOCKS * 256) {
    const double v = (double)(g[e] * scale);
    s += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// clip by global norm (torch.nn.utils.clip_grad_norm_: coef = min(1, max/(norm+1e-6))),
// then Adam with L2 weight decay (torch.optim.Adam defaults, amsgrad off)
__global__ __launch_bounds__(256) void adam_clip_kernel(
    float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
    int64_t n, float grad_scale, float clip_max, const double *__restrict__ partial, float step_size,
    float bc2_sqrt, float b1, float b2, float eps, float wd, float *__restrict__ norm_out) {
  __shared__ float s_coef;
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < NORM_BLOCKS; ++k) t += partial[k];
    const float nrm = (float)sqrt(t);
    float coef = 1.0f;
    if (clip_max > 0.0f) coef = fminf(clip_max / (nrm + 1e-6f), 1.0f);
    s_coef = coef;
    if (norm_out && blockIdx.x == 0) norm_out[0] = nrm;
  }
  __syncthreads();
  const float coef = s_coef;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float gr = g[e] * grad_scale * coef;
    const float w = p[e];
    gr = gr + wd * w;
    const float mm = b1 * m[e] + (1.0f - b1) * gr;
    const float vv = b2 * v[e] + (1.0f - b2) * gr * gr;
    m[e] = mm;
    v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[e] = w - step_size * (mm / denom);
  }
}

}  // namespace gvae

using namespace gvae;

// Sampler: thresholded adjacency of generated graphs (reference experiments/sample.py:58-73).
// One work-group per sample.  A node exists when sigmoid(node logit) > th_node; the edge logit
// of a pair with a missing end point, and of the diagonal, is replaced by -10 before the
// sigmoid, exactly as the reference does, so the bit is sigmoid(value) > th_edge in every case.
__global__ __launch_bounds__(256) void sample_adj_kernel(const float *__restrict__ vec,
                                                         const float *__restrict__ node, int N,
                                                         float th_edge, float th_node,
                                                         uint8_t *__restrict__ adj,
                                                         uint8_t *__restrict__ exists,
                                                         int32_t *__restrict__ count) {
  __shared__ uint8_t ex[1024];
  __shared__ int s_cnt;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t U = (int64_t)N * (N - 1) / 2;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int mine = 0;
  for (int c = tid; c < N; c += 256) {
    const float p = 1.0f / (1.0f + expf(-node[(int64_t)g * N + c]));
    const uint8_t e = p > th_node ? 1 : 0;
    ex[c] = e;
    exists[(int64_t)g * N + c] = e;
    mine += e;
  }
  if (mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (tid == 0) count[g] = s_cnt;
  const float p_off = 1.0f / (1.0f + expf(10.0f));      // sigmoid(-10)
  const uint8_t bit_off = p_off > th_edge ? 1 : 0;
  const float *vg = vec + g * U;
  uint8_t *ag = adj + (int64_t)g * N * N;
  for (int e = tid; e < N * N; e += 256) {
    const int r = e / N, c = e - r * N;
    uint8_t b = bit_off;
    if (r != c && ex[r] && ex[c]) {
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      const int64_t pair = (int64_t)lo * N - (int64_t)lo * (lo + 1) / 2 + (hi - lo - 1);
      b = 1.0f / (1.0f + expf(-vg[pair])) > th_edge ? 1 : 0;
    }
    ag[e] = b;
  }
}


extern "C" {

int64_t gvae_param_layout(const gvae_dims *d, int64_t *offsets, int64_t *sizes) {
  if (!d || d->F <= 0 || d->H <= 0 || d->Z <= 0 || d->N < 2) return -1;
  return param_layout(d, offsets, sizes);
}

int64_t gvae_step_ws_bytes(const gvae_dims *d, int n_graphs) {
  if (!d || n_graphs <= 0) return 0;
  return carve(d, n_graphs, nullptr).bytes;
}

#define TRY(x)            \
  do {                    \
    int rc_ = (x);        \
    if (rc_) return rc_;  \
  } while (0)

// Encoder, heads, reparameterisation and decoder: everything of the forward that the
// backward needs (model.py:295-303).
static int forward_model(const gvae_dims *d, const float *d_params, float *d_bn_state,
                         const float *d_x, const float *d_adj, const int32_t *d_n_x,
                         const int32_t *d_R, int n_graphs, const float *d_eps, int train, void *d_ws,
                         hipStream_t s) {
  const int B = n_graphs, F = d->F, H = d->H, Z = d->Z, N = d->N, U = N * (N - 1) / 2;
  const Workspace w = carve(d, B, (char *)d_ws);
  const Params P = bind(d, d_params);
  float *rm1 = d_bn_state, *rv1 = rm1 + H, *rm2 = rv1 + H, *rv2 = rm2 + H;

  // encoder, pooling, heads, reparameterisation, first decoder layers (model.py:295-302): one
  // launch, a work-group per graph
  {
    ModelFwdArgs a;
    a.x = d_x; a.adj = d_adj; a.eps = d_eps; a.n_x = d_n_x; a.R = d_R;
    a.W1 = P.p[GVAE_P_CONV1_W]; a.b1 = P.p[GVAE_P_CONV1_B]; a.g1 = P.p[GVAE_P_BN1_W];
    a.be1 = P.p[GVAE_P_BN1_B]; a.rm1 = rm1; a.rv1 = rv1;
    a.W2 = P.p[GVAE_P_CONV2_W]; a.b2 = P.p[GVAE_P_CONV2_B]; a.g2 = P.p[GVAE_P_BN2_W];
    a.be2 = P.p[GVAE_P_BN2_B]; a.rm2 = rm2; a.rv2 = rv2;
    a.Wmu = P.p[GVAE_P_MU_W]; a.bmu = P.p[GVAE_P_MU_B]; a.Wlv = P.p[GVAE_P_LV_W]; a.blv = P.p[GVAE_P_LV_B];
    a.Wde1 = P.p[GVAE_P_DE1_W]; a.bde1 = P.p[GVAE_P_DE1_B]; a.Wdn1 = P.p[GVAE_P_DN1_W];
    a.bdn1 = P.p[GVAE_P_DN1_B];
    a.Ahat = w.Ahat; a.H0a = w.H0a; a.Y1 = w.Y1; a.H1 = w.H1; a.H0b = w.H0b; a.Y2 = w.Y2; a.H2 = w.H2;
    // fp64 activations of the any-size kernel: in buffers only the backward of this batch writes
    // (dAct + dY and dH0 .. dH0_2 are each at least 2 * B*N*H contiguous floats, 256-byte aligned)
    a.T64 = reinterpret_cast<double *>(w.dAct); a.Y64 = reinterpret_cast<double *>(w.dH0);
    a.mean1 = w.mean1; a.invstd1 = w.invstd1; a.varu1 = w.varu1;
    a.mean2 = w.mean2; a.invstd2 = w.invstd2; a.varu2 = w.varu2;
    a.gvec = w.gvec; a.mu = w.mu; a.lvraw = w.lvraw; a.z = w.z; a.he = w.he; a.hn = w.hn;
    a.bn_mode = w.bn_mode;
    a.N = N; a.F = F; a.H = H; a.Z = Z; a.train = train;
    TRY(launch_model_fwd_graph(a, B, s));
  }
  if (train) {
    BnRunningArgs r;
    r.mean[0] = w.mean1; r.var_unb[0] = w.varu1; r.run_mean[0] = rm1; r.run_var[0] = rv1;
    r.mean[1] = w.mean2; r.var_unb[1] = w.varu2; r.run_mean[1] = rm2; r.run_var[1] = rv2;
    r.layers = 2; r.B = B; r.H = H; r.momentum = 0.1f;
    TRY(launch_bn_running_update(r, s));
  }
  // second decoder layers (model.py:302-303): contractions over the whole batch
  {
    GemmDesc l2[2] = {
        desc_linear_fwd(w.he, H, P.p[GVAE_P_DE2_W], P.p[GVAE_P_DE2_B], w.vec, B, U, H, 0),
        desc_linear_fwd(w.hn, H, P.p[GVAE_P_DN2_W], P.p[GVAE_P_DN2_B], w.node, B, N, H, 0)};
    TRY(launch_gemm_group(l2, 2, w.splitk, w.splitk_floats, s));
  }
  return 0;
}

// Matching, assignment and the losses (model.py:305-373).  Reads the logits, mu and logvar the
// model part left in the workspace and writes only buffers the backward never touches, so it
// may run on another stream, concurrently with the backward of the same batch.
static int forward_match_loss(const gvae_dims *d, const float *d_adj, const int32_t *d_R,
                              int n_graphs, int r_max, const float *d_pos_weight, int loss_variant,
                              int matching, int mpm_iters, float tol, int early_stop, float *d_S,
                              const int64_t *d_S_off, const int32_t *h_chunk_begin, int n_chunks,
                              void *d_ws, float *d_out4, int64_t *d_perm, int32_t *d_iters_done,
                              int phases, hipStream_t s) {
  const int B = n_graphs, Z = d->Z, N = d->N;
  const Workspace w = carve(d, B, (char *)d_ws);
  const bool do_prep = phases & GVAE_PHASE_PREP, do_mpm = phases & GVAE_PHASE_MPM;
  const bool do_tail = phases & GVAE_PHASE_TAIL, do_assign = phases & GVAE_PHASE_ASSIGN_LOSS;
  // model_legacy.py:294-310 matches all N rows of the padded matrices (no cut to the real nodes)
  const bool indiv = loss_variant == GVAE_LOSS_INDIVIDUAL_RHO;
  const int32_t *d_R_loss = d_R;
  if (indiv) {
    if (do_prep)
      hipLaunchKernelGGL(fill_i32_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, w.r_full, B, N);
    d_R = w.r_full;
    r_max = N;
  }
  // A streamed matching whose similarity tensor fits the S buffer whole runs as three parts a
  // caller may enqueue separately (every call for a batch is told the same chunk list); a chunked
  // one walks its chunks through the buffer in the MPM part, each chunk with its own tail at once
  // (the chunks share the queue workspace)
  const bool one_chunk = n_chunks <= 1;
  const int64_t NN = (int64_t)N * N;
  // matching (model.py:312-323)
  if (matching == GVAE_MATCH_NONE) {
    if (do_prep) {
      hipLaunchKernelGGL(identity_perm_kernel, dim3(ceil_div(B * N, 256)), dim3(256), 0, s, d_perm, B, N);
      TRY(hipMemsetAsync(d_iters_done, 0, sizeof(int32_t) * B, s) == hipSuccess ? 0 : GVAE_ELAUNCH);
    }
  } else {
    if (do_prep) {
      TRY(launch_prob(w.vec, w.node, B, N, w.P, s));
      TRY(gvae_node_sim(d_adj, w.P, d_R, B, N, NN, w.NS, s));
      // (the streamed matching computes its first iteration from the constant x0 and writes X
      // itself: the uniform start is only read by the S-free kernel, or when nothing iterates)
      if (matching != GVAE_MATCH_STREAM || mpm_iters < 1)
        hipLaunchKernelGGL(x0_init_kernel, dim3(min(ceil_div(N * N, 256), 64), B), dim3(256), 0, s, d_R,
                           N, w.X);
      if (matching == GVAE_MATCH_STREAM && one_chunk)
        TRY(match_stream_prep(d_adj, w.P, w.NS, d_R, B, N, NN, r_max, GVAE_S_PADDED, d_S, d_S_off, w.X, N,
                              NN, mpm_iters, tol, early_stop, d_iters_done, w.mpm_ws, s));
    }
    if (do_mpm) {
      if (matching == GVAE_MATCH_STREAM && one_chunk) {
        TRY(match_stream_iterate(d_R, B, r_max, GVAE_S_PADDED, d_S, d_S_off, w.X, N, NN, mpm_iters, tol,
                                 early_stop, d_iters_done, w.mpm_ws, s));
      } else if (matching == GVAE_MATCH_STREAM) {
        // chunk by chunk through the one S buffer (stream order keeps a chunk's matching
        // ahead of the next chunk's overwrite)
        for (int c = 0; c < n_chunks; ++c) {
          const int g0 = h_chunk_begin[c], nb = h_chunk_begin[c + 1] - h_chunk_begin[c];
          TRY(match_stream_run(d_adj + g0 * NN, w.P + g0 * NN, w.NS + g0 * NN, d_R + g0, nb, N, NN,
                               r_max, GVAE_S_PADDED, d_S, d_S_off + g0, w.X + g0 * NN, N, NN, mpm_iters,
                               tol, early_stop, d_iters_done + g0, w.mpm_ws, true, s));
        }
      } else if (do_assign) {
        // S-free MPM with the assignment as its epilogue
        TRY(gvae_mpm_fused_assign(d_adj, w.P, w.NS, d_R, B, N, NN, r_max, w.X, N, NN, mpm_iters, tol,
                                  early_stop, d_iters_done, d_perm, N, w.status, s));
      } else {
        TRY(gvae_mpm_fused(d_adj, w.P, w.NS, d_R, B, N, NN, r_max, w.X, N, NN, mpm_iters, tol,
                           early_stop, d_iters_done, s));
      }
    }
    // The last iterations of the graphs the streaming launch handed over, by the S-free kernel.
    // As a part of its own it can run on another stream beside the NEXT batch's similarity tensor,
    // whose persistent launch the caller orders behind it: beside that launch the tail's kernels
    // (up to 1024 threads and 67 KiB of LDS per graph) find no room and would hold back this
    // batch's assignment until it ends.
    if (do_tail && matching == GVAE_MATCH_STREAM && one_chunk)
      TRY(match_stream_tail(d_adj, w.P, w.NS, d_R, B, N, NN, r_max, w.X, N, NN, mpm_iters, tol, early_stop,
                            d_iters_done, w.mpm_ws, s));
  }
  if (!do_assign) return 0;
  // assignment (model.py:319-323), unless the S-free MPM above has already done it
  if (matching == GVAE_MATCH_STREAM || (matching == GVAE_MATCH_FUSED && !do_mpm))
    TRY(gvae_lsap(w.X, N, (int64_t)N * N, d_R, B, r_max, d_perm, N, w.status, s));
  if (indiv) {
    TRY(launch_loss_fwd_indiv(w.vec, d_adj, w.mu, w.lvraw, d_perm, d_R_loss, B, N, Z, w.rec_i, w.kl_i,
                              w.maxlogit_i, w.pw_i, d_out4, s));
    TRY(hipMemcpyAsync(w.perm_fwd, d_perm, sizeof(int64_t) * (size_t)B * N, hipMemcpyDeviceToDevice,
                       s) == hipSuccess ? 0 : GVAE_ELAUNCH);
    return 0;
  }
  // losses (model.py:325-373)
  TRY(launch_loss_fwd(w.vec, w.node, d_adj, w.mu, w.lvraw, d_perm, d_R, B, N, Z, d_pos_weight,
                      loss_variant == GVAE_LOSS_DUMMY_MASKED, w.rec_i, w.kl_i, w.maxlogit_i, d_out4, s));
  return 0;
}

static int check_match_args(const gvae_dims *d, int n_graphs, int r_max, int loss_variant,
                            int matching, const float *d_S, const int64_t *d_S_off,
                            const int32_t *h_chunk_begin, int n_chunks, const char *who) {
  GVAE_REQUIRE(n_graphs > 0 && n_graphs <= 65535 && r_max > 0 && r_max <= d->N, GVAE_EINVAL,
               "%s: bad sizes (n_graphs=%d r_max=%d N=%d)", who, n_graphs, r_max, d->N);
  GVAE_REQUIRE(matching >= GVAE_MATCH_NONE && matching <= GVAE_MATCH_FUSED, GVAE_EINVAL,
               "%s: unknown matching mode %d", who, matching);
  GVAE_REQUIRE(loss_variant >= GVAE_LOSS_CURRENT && loss_variant <= GVAE_LOSS_INDIVIDUAL_RHO,
               GVAE_EINVAL, "%s: unknown loss variant %d", who, loss_variant);
  GVAE_REQUIRE(loss_variant != GVAE_LOSS_INDIVIDUAL_RHO || (r_max == d->N && matching != GVAE_MATCH_NONE),
               GVAE_EINVAL, "%s: the individual-rho loss matches all N=%d rows (r_max=%d) and needs a matching",
               who, d->N, r_max);
  GVAE_REQUIRE(matching != GVAE_MATCH_STREAM || (d_S && d_S_off), GVAE_EINVAL,
               "%s: streaming matching needs the S buffer and its offsets", who);
  GVAE_REQUIRE(n_chunks >= 0 && (n_chunks == 0 || h_chunk_begin), GVAE_EINVAL, "%s: bad chunk list",
               who);
  if (n_chunks > 0) {
    GVAE_REQUIRE(h_chunk_begin[0] == 0 && h_chunk_begin[n_chunks] == n_graphs, GVAE_EINVAL,
                 "%s: chunks must cover graphs 0..%d", who, n_graphs);
    for (int c = 0; c < n_chunks; ++c)
      GVAE_REQUIRE(h_chunk_begin[c] < h_chunk_begin[c + 1], GVAE_EINVAL,
                   "%s: chunk %d is empty or out of order", who, c);
  }
  return 0;
}

int gvae_step_forward(const gvae_dims *d, const float *d_params, float *d_bn_state, const float *d_x,
                      const float *d_adj, const int32_t *d_n_x, const int32_t *d_R, int n_graphs,
                      int r_max, const float *d_eps, const float *d_pos_weight, int loss_variant,
                      int train, int matching, int mpm_iters, float tol, int early_stop, float *d_S,
                      const int64_t *d_S_off, const int32_t *h_chunk_begin, int n_chunks,
                      void *d_ws, float *d_out4, int64_t *d_perm, int32_t *d_iters_done,
                      void *stream) {
  GVAE_REQUIRE(d && d_params && d_bn_state && d_x && d_adj && d_n_x && d_R && d_eps && d_pos_weight &&
                   d_ws && d_out4 && d_perm && d_iters_done,
               GVAE_EINVAL, "step_forward: null pointer");
  TRY(check_match_args(d, n_graphs, r_max, loss_variant, matching, d_S, d_S_off, h_chunk_begin,
                       n_chunks, "step_forward"));
  hipStream_t s = (hipStream_t)stream;
  TRY(forward_model(d, d_params, d_bn_state, d_x, d_adj, d_n_x, d_R, n_graphs, d_eps, train, d_ws, s));
  TRY(forward_match_loss(d, d_adj, d_R, n_graphs, r_max, d_pos_weight, loss_variant, matching,
                         mpm_iters, tol, early_stop, d_S, d_S_off, h_chunk_begin, n_chunks, d_ws,
                         d_out4, d_perm, d_iters_done, GVAE_PHASE_MATCH | GVAE_PHASE_ASSIGN_LOSS, s));
  return check_launch("step_forward");
}

int gvae_step_forward_model(const gvae_dims *d, const float *d_params, float *d_bn_state,
                            const float *d_x, const float *d_adj, const int32_t *d_n_x,
                            const int32_t *d_R, int n_graphs, const float *d_eps, int train,
                            void *d_ws, void *stream) {
  GVAE_REQUIRE(d && d_params && d_bn_state && d_x && d_adj && d_n_x && d_R && d_eps && d_ws,
               GVAE_EINVAL, "step_forward_model: null pointer");
  GVAE_REQUIRE(n_graphs > 0 && n_graphs <= 65535, GVAE_EINVAL,
               "step_forward_model: bad n_graphs=%d", n_graphs);
  TRY(forward_model(d, d_params, d_bn_state, d_x, d_adj, d_n_x, d_R, n_graphs, d_eps, train, d_ws,
                    (hipStream_t)stream));
  return check_launch("step_forward_model");
}

int gvae_step_forward_match_loss(const gvae_dims *d, const float *d_adj, const int32_t *d_R,
                                 int n_graphs, int r_max, const float *d_pos_weight,
                                 int loss_variant, int matching, int mpm_iters, float tol,
                                 int early_stop, float *d_S, const int64_t *d_S_off,
                                 const int32_t *h_chunk_begin, int n_chunks, void *d_ws,
                                 float *d_out4, int64_t *d_perm, int32_t *d_iters_done,
                                 int phases, void *stream) {
  GVAE_REQUIRE(d && d_adj && d_R && d_pos_weight && d_ws && d_out4 && d_perm && d_iters_done,
               GVAE_EINVAL, "step_forward_match_loss: null pointer");
  GVAE_REQUIRE(phases > 0 && !(phases & ~(GVAE_PHASE_MATCH | GVAE_PHASE_ASSIGN_LOSS)), GVAE_EINVAL,
               "step_forward_match_loss: phases=%d", phases);
  TRY(check_match_args(d, n_graphs, r_max, loss_variant, matching, d_S, d_S_off, h_chunk_begin,
                       n_chunks, "step_forward_match_loss"));
  TRY(forward_match_loss(d, d_adj, d_R, n_graphs, r_max, d_pos_weight, loss_variant, matching,
                         mpm_iters, tol, early_stop, d_S, d_S_off, h_chunk_begin, n_chunks, d_ws,
                         d_out4, d_perm, d_iters_done, phases, (hipStream_t)stream));
  return check_launch("step_forward_match_loss");
}

int gvae_step_backward(const gvae_dims *d, const float *d_params, const float *d_x,
                       const float *d_adj, const int32_t *d_n_x, const int32_t *d_R, int n_graphs,
                       const float *d_eps, const float *d_pos_weight, int loss_variant, float c_rec,
                       float c_kl, const float *d_up, void *d_ws, float *d_grads, void *stream) {
  GVAE_REQUIRE(d && d_params && d_x && d_adj && d_n_x && d_R && d_eps && d_pos_weight && d_ws &&
                   d_grads,
               GVAE_EINVAL, "step_backward: null pointer");
  GVAE_REQUIRE(n_graphs > 0 && n_graphs <= 65535, GVAE_EINVAL, "step_backward: bad n_graphs=%d",
               n_graphs);
  GVAE_REQUIRE(loss_variant >= GVAE_LOSS_CURRENT && loss_variant <= GVAE_LOSS_INDIVIDUAL_RHO,
               GVAE_EINVAL, "step_backward: unknown loss variant %d", loss_variant);
  hipStream_t s = (hipStream_t)stream;
  const int B = n_graphs, F = d->F, H = d->H, Z = d->Z, N = d->N, U = N * (N - 1) / 2;
  const Workspace w = carve(d, B, (char *)d_ws);
  const Params P = bind(d, d_params);
  int64_t off[GVAE_P_COUNT];
  param_layout(d, off, nullptr);
  auto G = [&](int i) { return d_grads + off[i]; };

  if (loss_variant == GVAE_LOSS_INDIVIDUAL_RHO)
    TRY(launch_loss_bwd_indiv(w.vec, d_adj, d_R, w.perm_fwd, w.pw_i, B, N, c_rec, d_up, w.dvec,
                              w.dnode, s));
  else
    TRY(launch_loss_bwd(w.vec, w.node, d_adj, d_R, B, N, d_pos_weight,
                        loss_variant == GVAE_LOSS_DUMMY_MASKED, c_rec, d_up, w.dvec, w.dnode, s));
  // edge and node heads, layer by layer: the four contractions of a layer are independent
  {
    GemmDesc l2[4] = {
        desc_linear_bwd_weight(w.dvec, w.he, H, G(GVAE_P_DE2_W), B, U, H, nullptr),
        desc_linear_bwd_input(w.dvec, P.p[GVAE_P_DE2_W], w.dhe, B, U, H, 0, nullptr),
        desc_linear_bwd_weight(w.dnode, w.hn, H, G(GVAE_P_DN2_W), B, N, H, nullptr),
        desc_linear_bwd_input(w.dnode, P.p[GVAE_P_DN2_W], w.dhn, B, N, H, 0, nullptr)};
    l2[1].relu_mask_act = w.he;   // backward of the hidden ReLUs, in the contraction's epilogue
    l2[3].relu_mask_act = w.hn;
    TRY(launch_gemm_group(l2, 4, w.splitk, w.splitk_floats, s));
  }
  if (model_chain_resident(N, H, Z)) {
    // the per-graph part of the chain in one launch, then everything that is summed over the
    // batch: six weight gradients as one group, twelve column sums as another
    ModelBwdArgs a;
    a.dhe = w.dhe; a.dhn = w.dhn;
    a.Wde1 = P.p[GVAE_P_DE1_W]; a.Wdn1 = P.p[GVAE_P_DN1_W]; a.Wmu = P.p[GVAE_P_MU_W];
    a.Wlv = P.p[GVAE_P_LV_W]; a.W2 = P.p[GVAE_P_CONV2_W];
    a.mu = w.mu; a.lvraw = w.lvraw; a.eps = d_eps; a.d_up = d_up;
    a.Ahat = w.Ahat; a.Y1 = w.Y1; a.H1 = w.H1; a.Y2 = w.Y2; a.H2 = w.H2;
    a.b1 = P.p[GVAE_P_CONV1_B]; a.g1 = P.p[GVAE_P_BN1_W]; a.mean1 = w.mean1; a.invstd1 = w.invstd1;
    a.b2 = P.p[GVAE_P_CONV2_B]; a.g2 = P.p[GVAE_P_BN2_W]; a.mean2 = w.mean2; a.invstd2 = w.invstd2;
    a.n_x = d_n_x; a.R = d_R; a.bn_mode = w.bn_mode;
    a.dz = w.dz; a.dmu = w.dmu; a.dlv = w.dlv; a.dgvec = w.dgvec; a.dY = w.dY;
    a.dH0_1 = w.dH0; a.dH0_2 = w.dH0_2;
    a.dgam1 = w.dgam_p; a.dbet1 = w.dbet_p; a.dbias1 = w.dbias_p;
    a.dgam2 = w.dgam_p2; a.dbet2 = w.dbet_p2; a.dbias2 = w.dbias_p2;
    a.ck = c_kl / (float)B;
    a.N = N; a.H = H; a.Z = Z;
    TRY(launch_model_bwd_graph(a, B, s));
    GemmDesc gw[6] = {
        desc_linear_bwd_weight(w.dhe, w.z, Z, G(GVAE_P_DE1_W), B, H, Z, nullptr),
        desc_linear_bwd_weight(w.dhn, w.z, Z, G(GVAE_P_DN1_W), B, H, Z, nullptr),
        desc_linear_bwd_weight(w.dmu, w.gvec, H, G(GVAE_P_MU_W), B, Z, H, nullptr),
        desc_linear_bwd_weight(w.dlv, w.gvec, H, G(GVAE_P_LV_W), B, Z, H, nullptr),
        desc_linear_bwd_weight(w.dH0_2, w.H1, H, G(GVAE_P_CONV2_W), B * N, H, H, nullptr),
        desc_linear_bwd_weight(w.dH0, d_x, F, G(GVAE_P_CONV1_W), B * N, H, F, nullptr)};
    TRY(launch_gemm_group(gw, 6, w.splitk, w.splitk_floats, s));
    ColsumGroup cs;
    cs.add(w.dvec, B, U, U, G(GVAE_P_DE2_B));
    cs.add(w.dnode, B, N, N, G(GVAE_P_DN2_B));
    cs.add(w.dhe, B, H, H, G(GVAE_P_DE1_B));
    cs.add(w.dhn, B, H, H, G(GVAE_P_DN1_B));
    cs.add(w.dmu, B, Z, Z, G(GVAE_P_MU_B));
    cs.add(w.dlv, B, Z, Z, G(GVAE_P_LV_B));
    cs.add(w.dgam_p2, B, H, H, G(GVAE_P_BN2_W));
    cs.add(w.dbet_p2, B, H, H, G(GVAE_P_BN2_B));
    cs.add(w.dbias_p2, B, H, H, G(GVAE_P_CONV2_B));
    cs.add(w.dgam_p, B, H, H, G(GVAE_P_BN1_W));
    cs.add(w.dbet_p, B, H, H, G(GVAE_P_BN1_B));
    cs.add(w.dbias_p, B, H, H, G(GVAE_P_CONV1_B));
    TRY(launch_colsum_group(cs, s));
    return check_launch("step_backward");
  }
  {
    ColsumGroup cs;   // the four bias gradients of the two heads
    cs.add(w.dvec, B, U, U, G(GVAE_P_DE2_B));
    cs.add(w.dnode, B, N, N, G(GVAE_P_DN2_B));
    cs.add(w.dhe, B, H, H, G(GVAE_P_DE1_B));
    cs.add(w.dhn, B, H, H, G(GVAE_P_DN1_B));
    TRY(launch_colsum_group(cs, s));
  }
  {
    GemmDesc l1[3] = {
        desc_linear_bwd_weight(w.dhe, w.z, Z, G(GVAE_P_DE1_W), B, H, Z, nullptr),
        desc_linear_bwd_weight(w.dhn, w.z, Z, G(GVAE_P_DN1_W), B, H, Z, nullptr),
        desc_linear_bwd_input(w.dhe, P.p[GVAE_P_DE1_W], w.dz, B, H, Z, 0, nullptr)};
    TRY(launch_gemm_group(l1, 3, w.splitk, w.splitk_floats, s));
  }
  // dz = edge-head term + node-head term, in this order
  TRY(linear_bwd_input(w.dhn, P.p[GVAE_P_DN1_W], w.dz, B, H, Z, 1, w.splitk, s));
  // reparameterisation + KL
  TRY(launch_reparam_bwd(w.dz, w.mu, w.lvraw, d_eps, B, Z, c_kl, d_up, w.dmu, w.dlv, s));
  {
    GemmDesc hd[3] = {
        desc_linear_bwd_weight(w.dmu, w.gvec, H, G(GVAE_P_MU_W), B, Z, H, nullptr),
        desc_linear_bwd_weight(w.dlv, w.gvec, H, G(GVAE_P_LV_W), B, Z, H, nullptr),
        desc_linear_bwd_input(w.dmu, P.p[GVAE_P_MU_W], w.dgvec, B, Z, H, 0, nullptr)};
    TRY(launch_gemm_group(hd, 3, w.splitk, w.splitk_floats, s));
  }
  {
    ColsumGroup cs;
    cs.add(w.dmu, B, Z, Z, G(GVAE_P_MU_B));
    cs.add(w.dlv, B, Z, Z, G(GVAE_P_LV_B));
    TRY(launch_colsum_group(cs, s));
  }
  TRY(linear_bwd_input(w.dlv, P.p[GVAE_P_LV_W], w.dgvec, B, Z, H, 1, w.splitk, s));
  // encoder layer 2
  TRY(launch_pool_bwd(w.dgvec, d_n_x, d_R, B, N, H, w.dAct, s));
  TRY(launch_bn_relu_bwd(w.dAct, w.Y2, P.p[GVAE_P_CONV2_B], w.H2, P.p[GVAE_P_BN2_W], w.mean2,
                         w.invstd2, d_n_x, B, N, H, w.bn_mode, w.dY, w.dgam_p, w.dbet_p, w.dbias_p, s));
  {
    ColsumGroup cs;
    cs.add(w.dgam_p, B, H, H, G(GVAE_P_BN2_W));
    cs.add(w.dbet_p, B, H, H, G(GVAE_P_BN2_B));
    cs.add(w.dbias_p, B, H, H, G(GVAE_P_CONV2_B));
    TRY(launch_colsum_group(cs, s));
  }
  TRY(aggregate(w.Ahat, w.dY, w.dH0, B, N, H, s));  // Ahat is symmetric
  {
    GemmDesc c2[2] = {
        desc_linear_bwd_weight(w.dH0, w.H1, H, G(GVAE_P_CONV2_W), B * N, H, H, nullptr),
        desc_linear_bwd_input(w.dH0, P.p[GVAE_P_CONV2_W], w.dAct, B * N, H, H, 0, nullptr)};
    TRY(launch_gemm_group(c2, 2, w.splitk, w.splitk_floats, s));
  }
  // encoder layer 1
  TRY(launch_bn_relu_bwd(w.dAct, w.Y1, P.p[GVAE_P_CONV1_B], w.H1, P.p[GVAE_P_BN1_W], w.mean1,
                         w.invstd1, d_n_x, B, N, H, w.bn_mode, w.dY, w.dgam_p, w.dbet_p, w.dbias_p, s));
  {
    ColsumGroup cs;
    cs.add(w.dgam_p, B, H, H, G(GVAE_P_BN1_W));
    cs.add(w.dbet_p, B, H, H, G(GVAE_P_BN1_B));
    cs.add(w.dbias_p, B, H, H, G(GVAE_P_CONV1_B));
    TRY(launch_colsum_group(cs, s));
  }
  TRY(aggregate(w.Ahat, w.dY, w.dH0, B, N, H, s));
  TRY(linear_bwd_weight(w.dH0, d_x, F, G(GVAE_P_CONV1_W), B * N, H, F, w.splitk, s));
  return check_launch("step_backward");
}

int gvae_step_read_debug(const gvae_dims *d, int n_graphs, void *d_ws, int which, float **out_ptr,
                         int64_t *out_count) {
  GVAE_REQUIRE(d && d_ws && out_ptr && out_count && n_graphs > 0, GVAE_EINVAL, "read_debug: bad args");
  const Workspace w = carve(d, n_graphs, (char *)d_ws);
  const int64_t B = n_graphs, N = d->N, Z = d->Z, U = N * (N - 1) / 2;
  switch (which) {
    case 0: *out_ptr = w.vec; *out_count = B * U; break;
    case 1: *out_ptr = w.node; *out_count = B * N; break;
    case 2: *out_ptr = w.P; *out_count = B * N * N; break;
    case 3: *out_ptr = w.X; *out_count = B * N * N; break;
    case 4: *out_ptr = w.mu; *out_count = B * Z; break;
    case 5: *out_ptr = w.lvraw; *out_count = B * Z; break;
    case 6: *out_ptr = w.rec_i; *out_count = B; break;
    case 7: *out_ptr = w.kl_i; *out_count = B; break;
    case 8: *out_ptr = reinterpret_cast<float *>(w.status); *out_count = B; break;
    case 9: *out_ptr = w.Y1; *out_count = B * N * d->H; break;
    case 10: *out_ptr = w.H1; *out_count = B * N * d->H; break;
    case 11: *out_ptr = w.Y2; *out_count = B * N * d->H; break;
    case 12: *out_ptr = w.H2; *out_count = B * N * d->H; break;
    case 13: *out_ptr = w.gvec; *out_count = B * d->H; break;
    case 14: *out_ptr = w.Ahat; *out_count = B * N * N; break;
    case 15: *out_ptr = w.dY; *out_count = B * N * d->H; break;
    case 16: *out_ptr = w.dgvec; *out_count = B * d->H; break;
    case 17: *out_ptr = reinterpret_cast<float *>(w.mpm_ws);
             *out_count = gvae_mpm_stream_ws_bytes(n_graphs, d->N) / 4; break;
    default: return set_error(GVAE_EINVAL, "read_debug: unknown buffer %d", which);
  }
  return GVAE_OK;
}

static inline int64_t pad4f(int64_t x) { return (x + 3) & ~3LL; }

int64_t gvae_sample_ws_bytes(const gvae_dims *d, int n) {
  if (!d || n <= 0) return 0;
  return 4 * 2 * pad4f((int64_t)n * d->H);
}

int gvae_sample_decode(const gvae_dims *d, const float *d_params, const float *d_z, int n,
                       float th_edge, float th_node, float *d_edge_logits, float *d_node_logits,
                       uint8_t *d_adj, uint8_t *d_exists, int32_t *d_count, void *d_ws,
                       void *stream) {
  GVAE_REQUIRE(d && d_params && d_z && d_edge_logits && d_node_logits && d_adj && d_exists &&
                   d_count && d_ws,
               GVAE_EINVAL, "sample_decode: null pointer");
  GVAE_REQUIRE(n > 0 && n <= 65535 && d->N >= 2 && d->N <= 1024, GVAE_EINVAL,
               "sample_decode: bad sizes (n=%d N=%d)", n, d->N);
  hipStream_t s = (hipStream_t)stream;
  const int H = d->H, Z = d->Z, N = d->N;
  const int U = N * (N - 1) / 2;
  const Params P = bind(d, d_params);
  float *he = (float *)d_ws, *hn = he + pad4f((int64_t)n * H);
  TRY(linear_fwd(d_z, Z, P.p[GVAE_P_DE1_W], P.p[GVAE_P_DE1_B], he, n, H, Z, 1, s));
  TRY(linear_fwd(he, H, P.p[GVAE_P_DE2_W], P.p[GVAE_P_DE2_B], d_edge_logits, n, U, H, 0, s));
  TRY(linear_fwd(d_z, Z, P.p[GVAE_P_DN1_W], P.p[GVAE_P_DN1_B], hn, n, H, Z, 1, s));
  TRY(linear_fwd(hn, H, P.p[GVAE_P_DN2_W], P.p[GVAE_P_DN2_B], d_node_logits, n, N, H, 0, s));
  hipLaunchKernelGGL(sample_adj_kernel, dim3(n), dim3(256), 0, s, d_edge_logits, d_node_logits, N,
                     th_edge, th_node, d_adj, d_exists, d_count);
  return check_launch("sample_decode");
}

int gvae_step_accumulate_loss(const float *d_out4, float beta, float *d_out4_copy, double *d_loss_sum,
                              void *stream) {
  GVAE_REQUIRE(d_out4 && d_loss_sum, GVAE_EINVAL, "step_accumulate_loss: null pointer");
  hipLaunchKernelGGL(accumulate_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out4, beta,
                     d_out4_copy, d_loss_sum);
  return check_launch("step_accumulate_loss");
}

int64_t gvae_adam_ws_bytes(void) { return NORM_BLOCKS * sizeof(double); }

int gvae_adam_step(float *d_params, const float *d_grads, float *d_m, float *d_v, int64_t n,
                   float grad_scale, float clip_max, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, void *d_ws, float *d_norm_out, void *stream) {
  GVAE_REQUIRE(d_params && d_grads && d_m && d_v && d_ws && n > 0 && step >= 1, GVAE_EINVAL,
               "adam_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  double *partial = (double *)d_ws;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, d_grads, n, grad_scale,
                     partial);
  const int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks)),
                     dim3(256), 0, s, d_params, d_grads, d_m, d_v, n, grad_scale, clip_max, partial,
                     step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, d_norm_out);
  return check_launch("adam_step");
}

}  // extern "C"
