// Dense side of the GraphVAE step on gfx950: an fp32 matrix-core GEMM used for
// every encoder/decoder contraction (forward and backward), and the small fused
// kernels around it (GCN normalisation, per-graph BatchNorm+ReLU, pooling,
// reparameterisation, probability matrix, losses).
//
// Reference semantics: graphvae/models/model.py:295-303 (encoder, heads, decoder),
// :305-307,315 (dense logits -> probabilities), :325-361 (losses).  GCNConv follows
// PyG 2.6.1 defaults (self loops on every feature row, D^-1/2 (A+I) D^-1/2,
// linear without bias -> aggregate -> + bias).
#include "gvae_dense.h"

namespace gvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// GEMM: 64x64 block tile, 4 wavefronts, each wavefront one 32x32 accumulator tile
// fed by v_mfma_f32_32x32x2_f32; K staged through LDS 64 at a time.
// Operand lane map (32x32x2): lane l supplies A[m = l&31][k = l>>5] and
// B[k = l>>5][n = l&31]; accumulator register r of lane l is
// C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
// ---------------------------------------------------------------------------
#define GT_M 64
#define GT_N 64
#define GT_K 64
#define GT_LD 68

// These GEMMs are small and latency-bound (a few K-tiles per work-group), so a tile's global
// loads (16 + 16 scalars per lane) are all issued back to back: one memory round trip per
// 64-deep K-tile; long K is cut with split-K on the host side so that the chip is filled.
__device__ __forceinline__ void gemm_tile(const GemmDesc &d, int bx, int by, int z) {
  __shared__ float As[GT_K][GT_LD];
  __shared__ float Bs[GT_K][GT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int m0 = by * GT_M, n0 = bx * GT_N;
  const float *A = d.A, *B = d.B;
  float *C = d.C;
  int k_begin = 0, k_end = d.K;
  if (d.splits > 1) {
    const int chunk = ((d.K + d.splits - 1) / d.splits + GT_K - 1) / GT_K * GT_K;
    k_begin = z * chunk;
    k_end = min(d.K, k_begin + chunk);
    C = d.splitk_ws + (int64_t)z * d.M * d.N;
  } else {
    A += z * d.a_bs;
    B += z * d.b_bs;
    C += z * d.c_bs;
  }
  f32x16 acc = {0};
  const bool a_kfast = (d.a_cs == 1), b_nfast = (d.b_cs == 1);
  // lane -> (row, k) of the A tile and (k, col) of the B tile; the fast-varying index follows
  // the operand's unit stride so that a wave's loads are contiguous
  const int am = a_kfast ? (tid >> 6) : (tid & 63), ak = a_kfast ? (tid & 63) : (tid >> 6);
  const int bn = b_nfast ? (tid & 63) : (tid >> 6), bk = b_nfast ? (tid >> 6) : (tid & 63);
  // Addresses = uniform pointer (scalar registers: tile origin + r * uniform step) + ONE 32-bit
  // byte offset per lane and operand (operands are far below 4 GiB; checked at launch).  With a
  // 64-bit address per load the kernel needed 185 registers; at 128 four work-groups fit a CU
  // and the kernel finds room beside one that leaves a quarter of the register file free.
  const unsigned a_rs = (unsigned)d.a_rs * 4u, a_cs = (unsigned)d.a_cs * 4u;
  const unsigned b_rs = (unsigned)d.b_rs * 4u, b_cs = (unsigned)d.b_cs * 4u;
  const unsigned a_lane = (unsigned)(m0 + am) * a_rs + (unsigned)ak * a_cs;
  const unsigned b_lane = (unsigned)bk * b_rs + (unsigned)(n0 + bn) * b_cs;
  const unsigned a_step = 4u * (a_kfast ? a_rs : a_cs), b_step = 4u * (b_nfast ? b_rs : b_cs);
  const char *Ab = reinterpret_cast<const char *>(A), *Bb = reinterpret_cast<const char *>(B);
  for (int k0 = k_begin; k0 < k_end; k0 += GT_K) {
    float ra[16], rb[16];
    const char *Ak = Ab + (size_t)k0 * a_cs, *Bk = Bb + (size_t)k0 * b_rs;     // uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = a_kfast ? am + 4 * r : am, k = a_kfast ? ak : ak + 4 * r;
      const int gm = m0 + m, gk = k0 + k;
      ra[r] = (gm < d.M && gk < k_end)
                  ? *reinterpret_cast<const float *>(Ak + (size_t)r * a_step + a_lane)
                  : 0.0f;
      const int n = b_nfast ? bn : bn + 4 * r, kb = b_nfast ? bk + 4 * r : bk;
      const int gn = n0 + n, gkb = k0 + kb;
      rb[r] = (gn < d.N && gkb < k_end)
                  ? *reinterpret_cast<const float *>(Bk + (size_t)r * b_step + b_lane)
                  : 0.0f;
    }
    __syncthreads();  // previous tile's MFMA reads are done
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = a_kfast ? am + 4 * r : am, k = a_kfast ? ak : ak + 4 * r;
      As[k][m] = ra[r];
      const int n = b_nfast ? bn : bn + 4 * r, kb = b_nfast ? bk + 4 * r : bk;
      Bs[kb][n] = rb[r];
    }
    __syncthreads();
    const int kmax = min(GT_K, k_end - k0);
    for (int kk = 0; kk < kmax; kk += 2) {
      const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  const int col = n0 + wn + (lane & 31);
  if (col >= d.N) return;
  const bool direct = d.splits <= 1;
  const float bv = (direct && d.bias) ? d.bias[col] : 0.0f;
  const int64_t crs = direct ? d.c_rs : d.N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row >= d.M) continue;
    float v = acc[r];
    if (direct) {
      v += bv;
      if (d.accumulate) v += C[row * crs + col];
      if (d.relu) v = fmaxf(v, 0.0f);
      if (d.relu_mask_act && !(d.relu_mask_act[row * crs + col] > 0.0f)) v = 0.0f;
    }
    C[row * crs + col] = v;
  }
}

__global__ __launch_bounds__(256, 4) void gemm_kernel(GemmDesc d) {
  gemm_tile(d, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent GEMMs in one launch.  The step's contractions are small (a few to a few
// dozen tiles each), so members that do not depend on each other fill the chip together
// instead of one after the other.  The grid is one-dimensional with exactly one work-group per
// tile of every member (a bounding box over members of different shapes -- a split-K weight
// gradient of 4 tiles x 64 splits beside an input gradient of 256 tiles -- dispatched 16 640
// work-groups of which 16 000 left at once: 43 us for 10 us of work).
__global__ __launch_bounds__(256, 4) void gemm_group_kernel(GemmGroup g) {
  int k = 0;
  while (k + 1 < g.n && (int)blockIdx.x >= g.wg_begin[k + 1]) ++k;
  const int local = (int)blockIdx.x - g.wg_begin[k];
  const int per_z = g.tiles_x[k] * g.tiles_y[k];
  const int z = local / per_z, rem = local - z * per_z;
  const int by = rem / g.tiles_x[k], bx = rem - by * g.tiles_x[k];
  gemm_tile(g.d[k], bx, by, z);
}

__device__ __forceinline__ void splitk_reduce(const GemmDesc &d, int64_t e) {
  if (e >= (int64_t)d.M * d.N) return;
  const int row = e / d.N, col = e - (int64_t)row * d.N;
  float v = 0.0f;
  for (int z = 0; z < d.splits; ++z) v += d.splitk_ws[(int64_t)z * d.M * d.N + e];
  if (d.bias) v += d.bias[col];
  if (d.accumulate) v += d.C[row * d.c_rs + col];
  if (d.relu) v = fmaxf(v, 0.0f);
  if (d.relu_mask_act && !(d.relu_mask_act[row * d.c_rs + col] > 0.0f)) v = 0.0f;
  d.C[row * d.c_rs + col] = v;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmDesc d) {
  splitk_reduce(d, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(GemmGroup g) {
  int k = 0;
  while (k + 1 < g.n && (int)blockIdx.x >= g.red_begin[k + 1]) ++k;
  splitk_reduce(g.d[k], (int64_t)((int)blockIdx.x - g.red_begin[k]) * 256 + threadIdx.x);
}

// the kernel addresses an operand as base + 32-bit byte offset (one batch member at a time)
static bool operands_below_4gib(const GemmDesc &d) {
  const int64_t a = ((int64_t)(d.M + GT_M) * d.a_rs + (int64_t)(d.K + GT_K) * d.a_cs) * 4;
  const int64_t b = ((int64_t)(d.K + GT_K) * d.b_rs + (int64_t)(d.N + GT_N) * d.b_cs) * 4;
  return d.a_rs >= 0 && d.a_cs >= 0 && d.b_rs >= 0 && d.b_cs >= 0 && a < (1LL << 32) && b < (1LL << 32);
}

int launch_gemm(const GemmDesc &d, hipStream_t s) {
  GVAE_REQUIRE(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, GVAE_EINVAL,
               "gemm: bad descriptor (M=%d N=%d K=%d batch=%d)", d.M, d.N, d.K, d.batch);
  GVAE_REQUIRE(operands_below_4gib(d), GVAE_ELIMIT, "gemm: operand of 4 GiB or more (M=%d N=%d K=%d)",
               d.M, d.N, d.K);
  GVAE_REQUIRE(d.splits == 1 || (d.batch == 1 && d.splitk_ws), GVAE_EINVAL,
               "gemm: split-K needs batch == 1 and a workspace");
  dim3 grid(ceil_div(d.N, GT_N), ceil_div(d.M, GT_M), d.splits > 1 ? d.splits : d.batch);
  GVAE_REQUIRE(grid.y <= 65535 && grid.z <= 65535, GVAE_ELIMIT, "gemm: grid too large");
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, s, d);
  if (d.splits > 1) {
    const int64_t n = (int64_t)d.M * d.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
  }
  return check_launch("gemm");
}

int launch_gemm_group(GemmDesc *descs, int n, float *splitk_base, int64_t splitk_floats,
                      hipStream_t s) {
  GVAE_REQUIRE(descs && n >= 1 && n <= GEMM_GROUP_MAX, GVAE_EINVAL, "gemm group: bad size %d", n);
  // every split member gets its own stretch of the split-K workspace; a group that does not
  // fit (or has a single member) runs as individual launches
  int64_t used = 0;
  bool fits = n > 1;
  for (int k = 0; k < n; ++k) {
    if (descs[k].splits > 1) {
      descs[k].splitk_ws = splitk_base + used;
      used += (((int64_t)descs[k].splits * descs[k].M * descs[k].N) + 63) & ~63LL;
    }
  }
  if (used > splitk_floats) fits = false;
  if (!fits) {
    for (int k = 0; k < n; ++k) {
      if (descs[k].splits > 1) descs[k].splitk_ws = splitk_base;
      const int rc = launch_gemm(descs[k], s);
      if (rc) return rc;
    }
    return GVAE_OK;
  }
  GemmGroup g;
  g.n = n;
  int64_t n_wg = 0, n_red = 0;
  for (int k = 0; k < n; ++k) {
    const GemmDesc &d = descs[k];
    GVAE_REQUIRE(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, GVAE_EINVAL,
                 "gemm group: bad descriptor %d (M=%d N=%d K=%d batch=%d)", k, d.M, d.N, d.K, d.batch);
    GVAE_REQUIRE(d.splits == 1 || d.batch == 1, GVAE_EINVAL, "gemm group: split-K needs batch == 1");
    GVAE_REQUIRE(operands_below_4gib(d), GVAE_ELIMIT,
                 "gemm group: operand of 4 GiB or more (M=%d N=%d K=%d)", d.M, d.N, d.K);
    g.d[k] = d;
    g.tiles_x[k] = ceil_div(d.N, GT_N);
    g.tiles_y[k] = ceil_div(d.M, GT_M);
    g.wg_begin[k] = (int)n_wg;
    g.red_begin[k] = (int)n_red;
    n_wg += (int64_t)g.tiles_x[k] * g.tiles_y[k] * (d.splits > 1 ? d.splits : d.batch);
    if (d.splits > 1) n_red += ((int64_t)d.M * d.N + 255) / 256;
    GVAE_REQUIRE(n_wg < (1LL << 30) && n_red < (1LL << 30), GVAE_ELIMIT, "gemm group: grid too large");
  }
  for (int k = n; k <= GEMM_GROUP_MAX; ++k) {
    g.wg_begin[k] = (int)n_wg;
    g.red_begin[k] = (int)n_red;
  }
  hipLaunchKernelGGL(gemm_group_kernel, dim3((unsigned)n_wg), dim3(256), 0, s, g);
  if (n_red > 0)
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)n_red), dim3(256), 0, s, g);
  return check_launch("gemm group");
}

// ---------------------------------------------------------------------------
// Per-graph BatchNorm1d (eps 1e-5) + ReLU over the graph's n_x live rows: running statistics and
// the backward (one work-group per (graph, 64-column slab), 4 row lanes per column); the forward
// is a stage of the model kernels further down.
// ---------------------------------------------------------------------------

// Column sums are accumulated in double (as PyTorch's CPU BatchNorm does: its accumulation type
// for float is double) by four row lanes per column and folded in a fixed order.
__device__ __forceinline__ double quad_rows_sum(double v, double (*sh)[64], int rl, int c) {
  sh[rl][c] = v;
  __syncthreads();
  const double t = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
  __syncthreads();
  return t;
}

// The reference applies BatchNorm graph by graph (model.py:295-296), i.e. B sequential
// momentum updates r <- (1-m) r + m s_g in batch order.  Their closed form
//   r_B = (1-m)^B r_0 + sum_g m (1-m)^(B-1-g) s_g
// is evaluated here with one work-group per column (sum in double, one rounding).
__global__ __launch_bounds__(256) void bn_running_update_kernel(BnRunningArgs a) {
  __shared__ double sh[2][4];
  const int c = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const int B = a.B, H = a.H;
  const float *__restrict__ mean = a.mean[l], *__restrict__ var_unb = a.var_unb[l];
  const double keep = 1.0 - (double)a.momentum;
  double sm = 0.0, sv = 0.0;
  for (int g = tid; g < B; g += 256) {
    const double w = (double)a.momentum * pow(keep, (double)(B - 1 - g));
    sm += w * (double)mean[(int64_t)g * H + c];
    sv += w * (double)var_unb[(int64_t)g * H + c];
  }
  for (int off = 32; off > 0; off >>= 1) {
    sm += __shfl_xor(sm, off, 64);
    sv += __shfl_xor(sv, off, 64);
  }
  if ((tid & 63) == 0) { sh[0][tid >> 6] = sm; sh[1][tid >> 6] = sv; }
  __syncthreads();
  if (tid == 0) {
    const double k0 = pow(keep, (double)B);
    float *run_mean = a.run_mean[l], *run_var = a.run_var[l];
    run_mean[c] = (float)(k0 * (double)run_mean[c] + ((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3])));
    run_var[c] = (float)(k0 * (double)run_var[c] + ((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3])));
  }
}

int launch_bn_running_update(const BnRunningArgs &a, hipStream_t s) {
  GVAE_REQUIRE(a.layers >= 1 && a.layers <= 2, GVAE_EINVAL, "bn_running_update: %d layers", a.layers);
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(a.H, a.layers), dim3(256), 0, s, a);
  return check_launch("bn_running_update");
}

// backward of relu(BN(Y + bias)) (sums in double); the forward has left the CENTRED value
// y - mean = Y + bias - mean in the Y buffer (bias and mean are not read again):
//   dpre = dOut * (out > 0);  dbeta = sum dpre;  dot = sum dpre * (y - mean);  dgamma = dot * invstd
//   batch statistics (training forward):
//     dY = (dpre - dbeta/n - (y - mean) * dot * invstd^2 / n) * invstd * gamma
//   running statistics (eval forward; mean / invstd are constants of the input):
//     dY = dpre * invstd * gamma
//   rows >= n_x: 0.  mode[0] != 0: the forward of this batch used batch statistics.
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(
    const float *__restrict__ dOut, const float *__restrict__ Y, const float *__restrict__ bias,
    const float *__restrict__ out, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ invstd, const int32_t *__restrict__ n_x, int N, int H,
    const int32_t *__restrict__ mode, float *__restrict__ dY, float *__restrict__ dgamma_p,
    float *__restrict__ dbeta_p, float *__restrict__ dbias_p) {
  __shared__ double sh[4][64];
  const int g = blockIdx.y, tid = threadIdx.x;
  const int cl = tid & 63, rl = tid >> 6, c = blockIdx.x * 64 + cl;
  const bool cv = c < H;
  const int nx = n_x[g];
  const int64_t base = (int64_t)g * N * H;
  const float is = cv ? invstd[(int64_t)g * H + c] : 0.0f;
  double sb = 0.0, dot = 0.0;
  if (cv) for (int i = rl; i < nx; i += 4) {
    const int64_t e = base + (int64_t)i * H + c;
    const float dp = out[e] > 0.0f ? dOut[e] : 0.0f;
    sb += (double)dp;
    dot += (double)dp * (double)Y[e];
  }
  sb = quad_rows_sum(sb, sh, rl, cl);
  dot = quad_rows_sum(dot, sh, rl, cl);
  const float ga = cv ? gamma[c] : 0.0f;
  const bool batch_stats = mode[0] != 0;
  const float dmean = batch_stats ? (float)(sb / (double)nx) : 0.0f;
  const float k = batch_stats ? (float)(dot * (double)is * (double)is / (double)nx) : 0.0f;
  double sdy = 0.0;
  if (cv) for (int i = rl; i < N; i += 4) {
    const int64_t e = base + (int64_t)i * H + c;
    float v = 0.0f;
    if (i < nx) {
      const float dp = out[e] > 0.0f ? dOut[e] : 0.0f;
      v = (dp - dmean - Y[e] * k) * is * ga;
    }
    dY[e] = v;
    sdy += (double)v;
  }
  sdy = quad_rows_sum(sdy, sh, rl, cl);
  if (rl == 0 && cv) {
    dgamma_p[(int64_t)g * H + c] = (float)(dot * (double)is);
    dbeta_p[(int64_t)g * H + c] = (float)sb;
    dbias_p[(int64_t)g * H + c] = (float)sdy;
  }
}

int launch_bn_relu_bwd(const float *dOut, const float *Y, const float *bias, const float *out,
                       const float *gamma, const float *mean, const float *invstd,
                       const int32_t *n_x, int B, int N, int H, const int32_t *mode, float *dY,
                       float *dgamma_p, float *dbeta_p, float *dbias_p, hipStream_t s) {
  hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(ceil_div(H, 64), B), dim3(256), 0, s, dOut, Y, bias,
                     out, gamma, mean, invstd, n_x, N, H, mode, dY, dgamma_p, dbeta_p, dbias_p);
  return check_launch("bn_relu_bwd");
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ M, int rows, int cols,
                                                     int64_t ld, float *__restrict__ out,
                                                     int accumulate) {
  // 16 columns x 16 row lanes per work-group; each lane keeps 4 loads in flight
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int r = rl;
    for (; r + 48 < rows; r += 64) {
      s0 += M[(int64_t)r * ld + c];
      s1 += M[(int64_t)(r + 16) * ld + c];
      s2 += M[(int64_t)(r + 32) * ld + c];
      s3 += M[(int64_t)(r + 48) * ld + c];
    }
    for (; r < rows; r += 16) s0 += M[(int64_t)r * ld + c];
  }
  sh[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// several column sums in one launch (blockIdx.y = member), same arithmetic as colsum_kernel
__global__ __launch_bounds__(256) void colsum_group_kernel(ColsumGroup g) {
  __shared__ float sh[16][17];
  const int k = blockIdx.y;
  const float *__restrict__ M = g.M[k];
  const int rows = g.rows[k], cols = g.cols[k];
  const int64_t ld = g.ld[k];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  if ((int)blockIdx.x * 16 >= cols) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int r = rl;
    for (; r + 48 < rows; r += 64) {
      s0 += M[(int64_t)r * ld + c];
      s1 += M[(int64_t)(r + 16) * ld + c];
      s2 += M[(int64_t)(r + 32) * ld + c];
      s3 += M[(int64_t)(r + 48) * ld + c];
    }
    for (; r < rows; r += 16) s0 += M[(int64_t)r * ld + c];
  }
  sh[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][cl];
    g.out[k][c] = t;
  }
}

int launch_colsum_group(const ColsumGroup &g, hipStream_t s) {
  GVAE_REQUIRE(g.n >= 1 && g.n <= COLSUM_GROUP_MAX, GVAE_EINVAL, "colsum group: bad size %d", g.n);
  int max_cols = 0;
  for (int k = 0; k < g.n; ++k) max_cols = g.cols[k] > max_cols ? g.cols[k] : max_cols;
  hipLaunchKernelGGL(colsum_group_kernel, dim3(ceil_div(max_cols, 16), g.n), dim3(256), 0, s, g);
  return check_launch("colsum group");
}

int launch_colsum(const float *M, int rows, int cols, int64_t ld, float *out, int accumulate,
                  hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 16)), dim3(256), 0, s, M, rows, cols, ld, out,
                     accumulate);
  return check_launch("colsum");
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float *__restrict__ dgvec,
                                                       const int32_t *__restrict__ n_x,
                                                       const int32_t *__restrict__ Rarr, int N, int H,
                                                       float *__restrict__ dH) {
  const int g = blockIdx.y;
  const int nx = n_x[g];
  const float invR = 1.0f / (float)Rarr[g];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < N * H; e += gridDim.x * 256) {
    const int i = e / H, c = e - i * H;
    dH[(int64_t)g * N * H + e] = (i < nx) ? dgvec[(int64_t)g * H + c] * invR : 0.0f;
  }
}

int launch_pool_bwd(const float *dgvec, const int32_t *n_x, const int32_t *R, int B, int N, int H,
                    float *dH, hipStream_t s) {
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(min(ceil_div(N * H, 256), 64), B), dim3(256), 0, s, dgvec,
                     n_x, R, N, H, dH);
  return check_launch("pool_bwd");
}

// ---------------------------------------------------------------------------
// The model half of the forward for one graph per work-group: normalised adjacency, the two
// GCN layers (X W^T, Ahat * ., BatchNorm + ReLU), pooling, the two heads, reparameterisation and
// the first decoder layers, as stages of ONE kernel separated by work-group barriers.  Everything
// up to the pooled vector is local to a graph (BatchNorm runs over a graph's own rows), so the
// chain needs no grid-wide step.  Two kernels: model_fwd_lds_kernel (activations in LDS) and
// model_fwd_graph_kernel (any size, activations through global memory); both leave the tensors
// the backward reads in the workspace.
// ---------------------------------------------------------------------------
// out[n] = [relu](bias[n] + sum_k in[k] * W[n*K + k]) for one row, one lane per output
// (`in` was written earlier in the same kernel: no __restrict__, it must be read with ordinary
// vector loads after the barrier)
__device__ __forceinline__ void row_linear(const float *in, const float *__restrict__ W,
                                           const float *__restrict__ bias, float *__restrict__ out,
                                           int Nout, int K, int relu) {
  for (int n = threadIdx.x; n < Nout; n += blockDim.x) {
    const float *w = W + (int64_t)n * K;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) acc = __fmaf_rn(in[k], w[k], acc);
    acc += bias[n];
    out[n] = relu ? fmaxf(acc, 0.0f) : acc;
  }
}

// ---------------------------------------------------------------------------
// The chain for graphs of at most 128 nodes and a hidden width of at most 128 (a multiple of
// 32), with the graph's activations resident in LDS: every stage reads its operands from LDS (the
// weights from L2) and leaves its result there for the next one; the copies in the workspace,
// which the backward reads, are plain stores nobody waits for.  A stage then costs its arithmetic
// and a barrier instead of a round trip through global memory.
//
// The fp32 contractions of the backward run on the matrix cores, one 32x32 tile of the result per
// wavefront (v_mfma_f32_32x32x2_f32: lane l supplies A[row l&31][k l>>5] and B[k l>>5][col l&31]).
// Within a block of 8 consecutive k the lanes of half h hold k = 4h..4h+3, read as one 16-byte
// vector, and instruction t of the block contracts k = t and k = 4 + t: a fixed summation order
// like any other, with vector loads for both operands.  (The forward's are fp64, further down.)
// ---------------------------------------------------------------------------
struct FwdLds {
  int as, hs;                        // row strides (floats) of Ahat and of the activation buffers
  int o_ah, o_b, o_vec;              // offsets in floats; the first 2 KiB per 256 threads are doubles
  size_t bytes;
};
__host__ __device__ inline FwdLds fwd_lds_layout(int NP, int H, int Z) {
  FwdLds L;
  L.as = NP + 4;
  L.hs = H + 4;
  L.o_ah = (NP * 8 / 256) * 512;     // after double sh[NT/256][4][64]
  L.o_b = L.o_ah + NP * L.as;
  L.o_vec = L.o_b + NP * L.hs;       // dinv[NP], gvec[H], mu[Z], lv[Z], z[Z]
  L.bytes = (size_t)(L.o_vec + NP + H + 3 * Z) * sizeof(float);
  return L;
}

#define MFMA4(a, b)                                                      \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).x, (b).x, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).y, (b).y, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).z, (b).z, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).w, (b).w, acc, 0, 0, 0)

// C tile (rows m0.., cols n0..) = A (LDS, row stride as) * B, B(k, n) = Bl[k * bs + n] in LDS
__device__ __forceinline__ f32x16 tile_lds_lds(const float *A, int as, int m0, const float *Bl, int bs,
                                               int n0, int K) {
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const float *ap = A + (m0 + r) * as + 4 * half;
  const float *bp = Bl + (4 * half) * bs + n0 + r;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 8) {
    const float4 av = *reinterpret_cast<const float4 *>(ap + k0);
    float4 bv;
    bv.x = bp[(k0 + 0) * bs]; bv.y = bp[(k0 + 1) * bs];
    bv.z = bp[(k0 + 2) * bs]; bv.w = bp[(k0 + 3) * bs];
    MFMA4(av, bv);
  }
  return acc;
}

// accumulator register q of lane l is C[(q & 3) + 8 * (q >> 2) + 4 * (l >> 5)][l & 31]
__device__ __forceinline__ void tile_store(const f32x16 &acc, float *dst, int hs, int m0, int n0,
                                           float *__restrict__ glob, int N, int H) {
  const int lane = threadIdx.x & 63, col = n0 + (lane & 31);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = m0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    dst[row * hs + col] = acc[q];
    if (row < N) glob[(int64_t)row * H + col] = acc[q];
  }
}

// ---------------------------------------------------------------------------
// Forward in double precision.  BatchNorm over a handful of near-identical rows (a graph with
// interchangeable nodes plus its padded rows) divides by a standard deviation of the order of
// the fp32 rounding of its inputs: every fp32 rounding on the way to that division (of the
// normalised adjacency, of either contraction, of the layer-1 activations) then moves the
// gradients by 1e-5 .. 1e-4 of their scale.  The encoder forward of the LDS-resident kernel
// therefore runs in fp64 from the node features to the pooled vector (operands converted from
// the fp32 parameters, exact), on the matrix cores: v_mfma_f64_16x16x4_f64, lane l supplies
// A[row l&15][slot l>>4] and B[slot l>>4][col l&15], result register q of lane l is
// D[row (l>>4) + 4q][col l&15].  Slot s of instruction t contracts k = s*K/4 + t, so a lane's
// operands of consecutive instructions are consecutive in memory (vector loads); the summation
// order is fixed.  The normalised adjacency enters as dinv[i] * (mask * (dinv[k] * T[k])), with
// the 0/1 mask as bytes in LDS.  What the backward reads (centred pre-activations, activations,
// statistics, Ahat) is stored rounded to fp32, once.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
struct Acc64 {
  f64x4 c[2][2];    // 32x32 tile as 2x2 blocks of 16x16
};
struct Fwd64Lds {
  int asb, hs;                                  // bytes per mask row, doubles per activation row
  size_t o_b, o_a, o_dinv, o_vec, bytes;        // byte offsets
};
__host__ __device__ inline Fwd64Lds fwd64_lds_layout(int NP, int H, int Z) {
  Fwd64Lds L;
  L.asb = NP + 4;
  L.hs = H + 2;                                 // rows 16-byte aligned, consecutive rows 4 banks apart
  const size_t sh = (size_t)(NP * 8 / 256) * 2048;   // double sh[NT/256][4][64]
  L.o_b = sh < 4096 ? 4096 : sh;                // (the pooling keeps 4 partial rows of doubles there)
  L.o_a = L.o_b + (size_t)NP * L.hs * 8;
  L.o_dinv = L.o_a + (size_t)NP * L.asb;
  L.o_vec = L.o_dinv + (size_t)NP * 8;
  L.bytes = L.o_vec + (size_t)(H + 3 * Z) * sizeof(float);   // gvec[H], mu[Z], lv[Z], z[Z]
  return L;
}

__device__ __forceinline__ void mfma64_2x2(Acc64 &acc, double a0, double a1, double b0, double b1) {
  acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
  acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
  acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
  acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
}

// tile (rows m0.., cols n0..) = mask (bytes in LDS, row stride asb) * Bd (doubles in LDS), K % 16 == 0
__device__ __forceinline__ void tile64_mask_lds(Acc64 &acc, const uint8_t *M, int asb, int m0,
                                                const double *Bd, int hs, int n0, int K) {
  const int lane = threadIdx.x & 63, r = lane & 15, sl = lane >> 4, KQ = K >> 2;
  const uint8_t *ap0 = M + (m0 + r) * asb + sl * KQ, *ap1 = ap0 + 16 * asb;
  const double *bp = Bd + (sl * KQ) * hs + n0 + r;
  for (int t0 = 0; t0 < KQ; t0 += 4) {
    const uint32_t w0 = *reinterpret_cast<const uint32_t *>(ap0 + t0);
    const uint32_t w1 = *reinterpret_cast<const uint32_t *>(ap1 + t0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double a0 = (double)((w0 >> (8 * u)) & 0xffu), a1 = (double)((w1 >> (8 * u)) & 0xffu);
      const double *bq = bp + (t0 + u) * hs;
      mfma64_2x2(acc, a0, a1, bq[0], bq[16]);
    }
  }
}

// tile = Ad (doubles in LDS) * W^T, W (n, k) fp32 row-major in global memory with row stride K,
// K % 16 == 0, rows 16-byte aligned; a lane's next weights are requested before it multiplies
__device__ __forceinline__ void tile64_lds_w(Acc64 &acc, const double *Ad, int hs, int m0,
                                             const float *__restrict__ W, int n0, int K) {
  const int lane = threadIdx.x & 63, r = lane & 15, sl = lane >> 4, KQ = K >> 2;
  const double *ap0 = Ad + (m0 + r) * hs + sl * KQ, *ap1 = ap0 + 16 * hs;
  const float *wp0 = W + (n0 + r) * K + sl * KQ, *wp1 = wp0 + 16 * K;
  float4 w0 = *reinterpret_cast<const float4 *>(wp0), w1 = *reinterpret_cast<const float4 *>(wp1);
  for (int t0 = 0; t0 < KQ; t0 += 4) {
    const float4 c0 = w0, c1 = w1;
    if (t0 + 4 < KQ) {
      w0 = *reinterpret_cast<const float4 *>(wp0 + t0 + 4);
      w1 = *reinterpret_cast<const float4 *>(wp1 + t0 + 4);
    }
    const double2 a00 = *reinterpret_cast<const double2 *>(ap0 + t0);
    const double2 a01 = *reinterpret_cast<const double2 *>(ap0 + t0 + 2);
    const double2 a10 = *reinterpret_cast<const double2 *>(ap1 + t0);
    const double2 a11 = *reinterpret_cast<const double2 *>(ap1 + t0 + 2);
    mfma64_2x2(acc, a00.x, a10.x, (double)c0.x, (double)c1.x);
    mfma64_2x2(acc, a00.y, a10.y, (double)c0.y, (double)c1.y);
    mfma64_2x2(acc, a01.x, a11.x, (double)c0.z, (double)c1.z);
    mfma64_2x2(acc, a01.y, a11.y, (double)c0.w, (double)c1.w);
  }
}

template <class F>
__device__ __forceinline__ void tile64_foreach(const Acc64 &acc, int m0, int n0, F f) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        f(m0 + rb * 16 + (lane >> 4) + 4 * q, n0 + cb * 16 + (lane & 15), acc.c[rb][cb][q]);
}

// BatchNorm + ReLU of buf + bias in place (doubles in LDS), a 64-column slab per 256 threads, 4
// row lanes per column (quad_rows_sum); fp32 copies for the backward: the activation
// (out) and the centred pre-activation (yc)
template <int NT>
__device__ __forceinline__ void bn_relu_lds64(double *buf, int hs, int NP, int N, int H, int nx,
                                              int train, const float *__restrict__ bias,
                                              const float *__restrict__ gamma,
                                              const float *__restrict__ beta,
                                              const float *__restrict__ run_mean,
                                              const float *__restrict__ run_var,
                                              float *__restrict__ out, float *__restrict__ yc,
                                              float *__restrict__ mean_o, float *__restrict__ invstd_o,
                                              float *__restrict__ var_unb_o, double *shd) {
  const int tid = threadIdx.x, cl = tid & 63, rl = (tid >> 6) & 3, grp = tid >> 8;
  double(*sh)[64] = reinterpret_cast<double(*)[64]>(shd + grp * 256);
  for (int s0 = 0; s0 * 64 < H; s0 += NT / 256) {
    const int c = (s0 + grp) * 64 + cl;
    const bool cv = c < H;
    const double b = cv ? (double)bias[c] : 0.0;
    double mean, invstd;
    if (train) {
      double s = 0.0;
      if (cv) for (int i = rl; i < nx; i += 4) s += buf[i * hs + c] + b;
      s = quad_rows_sum(s, sh, rl, cl);
      mean = s / (double)nx;
      double q = 0.0;
      if (cv) for (int i = rl; i < nx; i += 4) {
        const double dlt = (buf[i * hs + c] + b) - mean;
        q += dlt * dlt;
      }
      q = quad_rows_sum(q, sh, rl, cl);
      invstd = 1.0 / sqrt(q / (double)nx + 1e-5);
      if (cv && rl == 0) {
        mean_o[c] = (float)mean;
        invstd_o[c] = (float)invstd;
        var_unb_o[c] = (float)(q / (double)(nx - 1));
      }
    } else {
      mean = cv ? (double)run_mean[c] : 0.0;
      invstd = cv ? 1.0 / sqrt((double)run_var[c] + 1e-5) : 0.0;
      if (cv && rl == 0) {
        mean_o[c] = (float)mean;
        invstd_o[c] = (float)invstd;
      }
    }
    if (cv) {
      const double ga = (double)gamma[c], be = (double)beta[c];
      for (int i = rl; i < NP; i += 4) {
        double o = 0.0, xc = 0.0;
        if (i < nx) {
          xc = (buf[i * hs + c] + b) - mean;
          o = fmax(xc * invstd * ga + be, 0.0);
        }
        buf[i * hs + c] = o;
        if (i < N) {
          out[(int64_t)i * H + c] = (float)o;
          yc[(int64_t)i * H + c] = (float)xc;
        }
      }
    }
  }
}

// The chain for any size (N <= 1024), a work-group per graph: the same fp64 arithmetic with plain fma
// chains, activations through two fp64 buffers in global memory (a work-group finds its own stores
// after a barrier).
__global__ __launch_bounds__(256) void model_fwd_graph_kernel(ModelFwdArgs a) {
  __shared__ double sh[4][64];
  __shared__ double dinv[1024];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int N = a.N, F = a.F, H = a.H, Z = a.Z;
  const int64_t NH = (int64_t)N * H;
  const int nx = a.n_x[g], R = a.R[g];
  const float *Ag = a.adj + (int64_t)g * N * N;
  // (no __restrict__: both are written and read again behind a barrier)
  double *T = a.T64 + g * NH, *Y = a.Y64 + g * NH;
  if (g == 0 && tid == 0) a.bn_mode[0] = a.train;   // which statistics: the backward reads it
  for (int i = tid; i < N; i += 256) {
    int cnt = 0;
    if (i < R)
      for (int j = 0; j < R; ++j) cnt += (j != i && Ag[(int64_t)i * N + j] != 0.0f) ? 1 : 0;
    dinv[i] = (i < nx) ? 1.0 / sqrt(1.0 + (double)cnt) : 0.0;
  }
  __syncthreads();
  auto edge = [&](int i, int k) -> bool {
    return i < nx && k < nx && (i == k || (i < R && k < R && Ag[(int64_t)i * N + k] != 0.0f));
  };
  {
    float *Og = a.Ahat + (int64_t)g * N * N;
    for (int e = tid; e < N * N; e += 256) {
      const int i = e / N, j = e - i * N;
      Og[e] = edge(i, j) ? (float)(dinv[i] * dinv[j]) : 0.0f;
    }
  }
  // buffer Y <- dinv[row] * sum over the row's neighbours k of T[k] (T holds dinv[k] * linear output)
  auto aggregate = [&]() {
    for (int e = tid; e < N * H; e += 256) {
      const int i = e / H, c = e - i * H;
      double acc = 0.0;
      if (i < nx)
        for (int k = 0; k < nx; ++k)
          if (edge(i, k)) acc += T[(int64_t)k * H + c];
      Y[e] = dinv[i] * acc;
    }
    __syncthreads();
  };
  // layer 1
  {
    const float *xg = a.x + (int64_t)g * N * F;
    for (int e = tid; e < N * H; e += 256) {
      const int i = e / H, c = e - i * H;
      double acc = 0.0;
      for (int f = 0; f < F; ++f) acc = fma((double)xg[(int64_t)i * F + f], (double)a.W1[(int64_t)c * F + f], acc);
      a.H0a[g * NH + e] = (float)acc;
      T[e] = acc * dinv[i];
    }
  }
  __syncthreads();
  aggregate();
  bn_relu_lds64<256>(Y, H, N, N, H, nx, a.train, a.b1, a.g1, a.be1, a.rm1, a.rv1, a.H1 + g * NH,
                     a.Y1 + g * NH, a.mean1 + (int64_t)g * H, a.invstd1 + (int64_t)g * H,
                     a.varu1 + (int64_t)g * H, &sh[0][0]);
  __syncthreads();
  // layer 2
  for (int e = tid; e < N * H; e += 256) {
    const int i = e / H, c = e - i * H;
    double acc = 0.0;
    for (int k = 0; k < H; ++k) acc = fma(Y[(int64_t)i * H + k], (double)a.W2[(int64_t)c * H + k], acc);
    a.H0b[g * NH + e] = (float)acc;
    T[e] = acc * dinv[i];
  }
  __syncthreads();
  aggregate();
  bn_relu_lds64<256>(Y, H, N, N, H, nx, a.train, a.b2, a.g2, a.be2, a.rm2, a.rv2, a.H2 + g * NH,
                     a.Y2 + g * NH, a.mean2 + (int64_t)g * H, a.invstd2 + (int64_t)g * H,
                     a.varu2 + (int64_t)g * H, &sh[0][0]);
  __syncthreads();
  for (int c = tid; c < H; c += 256) {
    double s0 = 0.0;
    for (int i = 0; i < nx; ++i) s0 += Y[(int64_t)i * H + c];
    a.gvec[(int64_t)g * H + c] = (float)(s0 / (double)R);
  }
  __syncthreads();
  // heads, reparameterisation, first decoder layers: one row per graph
  const float *gv = a.gvec + (int64_t)g * H;
  row_linear(gv, a.Wmu, a.bmu, a.mu + (int64_t)g * Z, Z, H, 0);
  row_linear(gv, a.Wlv, a.blv, a.lvraw + (int64_t)g * Z, Z, H, 0);
  __syncthreads();
  for (int k = threadIdx.x; k < Z; k += blockDim.x) {
    const int64_t e = (int64_t)g * Z + k;
    const float lv = fminf(fmaxf(a.lvraw[e], -4.0f), 4.0f);
    a.z[e] = a.mu[e] + a.eps[e] * expf(0.5f * lv);
  }
  __syncthreads();
  const float *zg = a.z + (int64_t)g * Z;
  row_linear(zg, a.Wde1, a.bde1, a.he + (int64_t)g * H, H, Z, 1);
  row_linear(zg, a.Wdn1, a.bdn1, a.hn + (int64_t)g * H, H, Z, 1);
}

// out[n] = [relu](bias[n] + sum_k in[k] * W[n*K + k]), in[] in LDS, K % 4 == 0, for the rows
// n = first, first + step, ... < Nout
__device__ __forceinline__ float row_dot4(const float *in, const float *__restrict__ w, int K) {
  float acc = 0.0f;
  for (int k = 0; k < K; k += 4) {
    const float4 wv = *reinterpret_cast<const float4 *>(w + k);
    const float4 xv = *reinterpret_cast<const float4 *>(in + k);
    acc = __fmaf_rn(xv.x, wv.x, acc);
    acc = __fmaf_rn(xv.y, wv.y, acc);
    acc = __fmaf_rn(xv.z, wv.z, acc);
    acc = __fmaf_rn(xv.w, wv.w, acc);
  }
  return acc;
}

template <int NP>
__global__ __launch_bounds__(NP * 8, 4) void model_fwd_lds_kernel(ModelFwdArgs a) {
  constexpr int NT = NP * 8;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const int N = a.N, F = a.F, H = a.H, Z = a.Z;
  const Fwd64Lds L = fwd64_lds_layout(NP, H, Z);
  const int ASB = L.asb, HS = L.hs;
  double *shd = reinterpret_cast<double *>(lds_raw);
  double *b = reinterpret_cast<double *>(lds_raw + L.o_b);
  uint8_t *Am = lds_raw + L.o_a;
  double *dinv = reinterpret_cast<double *>(lds_raw + L.o_dinv);
  float *gv = reinterpret_cast<float *>(lds_raw + L.o_vec), *mu_s = gv + H, *lv_s = mu_s + Z,
        *z_s = lv_s + Z;
  const int nx = a.n_x[g], R = a.R[g];
  const int64_t NH = (int64_t)N * H;
  if (g == 0 && tid == 0) a.bn_mode[0] = a.train;   // which statistics: the backward reads it

  // degrees: 8 lanes per row count the row's neighbours (exact in any order)
  {
    const int i = tid >> 3, sub = tid & 7;
    const float *Ag = a.adj + (int64_t)g * N * N;
    float cnt = 0.0f;
    if (i < R)
      for (int j = sub; j < R; j += 8)
        if (j != i) cnt += (Ag[(int64_t)i * N + j] != 0.0f) ? 1.0f : 0.0f;
    cnt += __shfl_xor(cnt, 1, 64);
    cnt += __shfl_xor(cnt, 2, 64);
    cnt += __shfl_xor(cnt, 4, 64);
    if (sub == 0) dinv[i] = (i < nx) ? 1.0 / sqrt(1.0 + (double)cnt) : 0.0;
  }
  __syncthreads();
  // X W1^T (K = F: plain fma chains), scaled by dinv[row] for the aggregation; rows >= n_x zero
  {
    const float *xg = a.x + (int64_t)g * N * F;
    for (int e = tid; e < NP * H; e += NT) {
      const int i = e / H, c = e - i * H;
      double acc = 0.0;
      if (i < N) {
        for (int f = 0; f < F; ++f) acc = fma((double)xg[i * F + f], (double)a.W1[c * F + f], acc);
        a.H0a[g * NH + e] = (float)acc;
      }
      b[i * HS + c] = acc * dinv[i];
    }
  }
  // adjacency mask with self loops on the n_x live rows; Ahat (fp32) for the backward
  {
    const float *Ag = a.adj + (int64_t)g * N * N;
    float *Og = a.Ahat + (int64_t)g * N * N;
    for (int e = tid; e < NP * NP; e += NT) {
      const int i = e / NP, j = e - i * NP;
      bool m = false;
      if (i < nx && j < nx) {
        if (i == j) m = true;
        else if (i < R && j < R) m = Ag[(int64_t)i * N + j] != 0.0f;
      }
      Am[i * ASB + j] = m ? 1 : 0;
      if (i < N && j < N) Og[(int64_t)i * N + j] = m ? (float)(dinv[i] * dinv[j]) : 0.0f;
    }
  }
  __syncthreads();
  // One activation buffer, updated in place: a contraction's tiles (at most one per wavefront:
  // (NP/32)(H/32) <= NP/8 wavefronts) are all computed before any is stored.
  const int tn = H / 32, tiles = (NP / 32) * tn;
  const bool has_tile = wave < tiles;
  const int m0 = (wave / tn) * 32, n0 = (wave % tn) * 32;
  Acc64 acc;
  auto zero = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc.c[u >> 1][u & 1] = f64x4{0.0, 0.0, 0.0, 0.0};
  };
  auto aggregate = [&]() {      // buffer <- dinv[row] * (mask * buffer)
    zero();
    if (has_tile) tile64_mask_lds(acc, Am, ASB, m0, b, HS, n0, NP);
    __syncthreads();
    if (has_tile) tile64_foreach(acc, m0, n0, [&](int row, int col, double v) { b[row * HS + col] = dinv[row] * v; });
    __syncthreads();
  };
  // layer 1: Y1 = Ahat * H0a;  H1 = relu(bn(Y1 + bias))
  aggregate();
  bn_relu_lds64<NT>(b, HS, NP, N, H, nx, a.train, a.b1, a.g1, a.be1, a.rm1, a.rv1, a.H1 + g * NH,
                    a.Y1 + g * NH, a.mean1 + (int64_t)g * H, a.invstd1 + (int64_t)g * H,
                    a.varu1 + (int64_t)g * H, shd);
  __syncthreads();
  // layer 2: H0b = H1 * W2^T;  Y2 = Ahat * H0b;  H2 = relu(bn(Y2 + bias))
  zero();
  if (has_tile) tile64_lds_w(acc, b, HS, m0, a.W2, n0, H);
  __syncthreads();
  if (has_tile) {
    float *glob = a.H0b + g * NH;
    tile64_foreach(acc, m0, n0, [&](int row, int col, double v) {
      b[row * HS + col] = dinv[row] * v;
      if (row < N) glob[(int64_t)row * H + col] = (float)v;
    });
  }
  __syncthreads();
  aggregate();
  bn_relu_lds64<NT>(b, HS, NP, N, H, nx, a.train, a.b2, a.g2, a.be2, a.rm2, a.rv2, a.H2 + g * NH,
                    a.Y2 + g * NH, a.mean2 + (int64_t)g * H, a.invstd2 + (int64_t)g * H,
                    a.varu2 + (int64_t)g * H, shd);
  __syncthreads();
  // pooled vector: four row lanes per column, then rounded to fp32 once
  {
    for (int e = tid; e < 4 * H; e += NT) {
      const int rl = e / H, c = e - rl * H;
      double s = 0.0;
      for (int i = rl; i < nx; i += 4) s += b[i * HS + c];
      shd[rl * 128 + c] = s;
    }
    __syncthreads();
    for (int c = tid; c < H; c += NT) {
      const float v = (float)(((shd[c] + shd[128 + c]) + (shd[256 + c] + shd[384 + c])) / (double)R);
      gv[c] = v;
      a.gvec[(int64_t)g * H + c] = v;
    }
  }
  __syncthreads();
  // heads
  for (int n = tid; n < 2 * Z; n += NT) {
    const bool lvh = n >= Z;
    const int o = lvh ? n - Z : n;
    const float v = row_dot4(gv, (lvh ? a.Wlv : a.Wmu) + (int64_t)o * H, H) + (lvh ? a.blv : a.bmu)[o];
    (lvh ? lv_s : mu_s)[o] = v;
    (lvh ? a.lvraw : a.mu)[(int64_t)g * Z + o] = v;
  }
  __syncthreads();
  for (int k = tid; k < Z; k += NT) {
    const float lv = fminf(fmaxf(lv_s[k], -4.0f), 4.0f);
    const float zz = mu_s[k] + a.eps[(int64_t)g * Z + k] * expf(0.5f * lv);
    z_s[k] = zz;
    a.z[(int64_t)g * Z + k] = zz;
  }
  __syncthreads();
  // first decoder layers
  for (int n = tid; n < 2 * H; n += NT) {
    const bool nh = n >= H;
    const int o = nh ? n - H : n;
    const float v = row_dot4(z_s, (nh ? a.Wdn1 : a.Wde1) + (int64_t)o * Z, Z) + (nh ? a.bdn1 : a.bde1)[o];
    (nh ? a.hn : a.he)[(int64_t)g * H + o] = fmaxf(v, 0.0f);
  }
}

int launch_model_fwd_graph(const ModelFwdArgs &a, int B, hipStream_t s) {
  GVAE_REQUIRE(a.N <= 1024, GVAE_ELIMIT, "model_fwd: N=%d > 1024", a.N);
  auto vec16 = [](const float *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // (the weight rows are read as 16-byte vectors: an odd number of pair logits can leave the
  // parameters behind them off that alignment)
  const bool resident = model_chain_resident(a.N, a.H, a.Z) && vec16(a.W2) && vec16(a.Wmu) && vec16(a.Wlv) && vec16(a.Wde1) && vec16(a.Wdn1);
  if (!resident) {
    hipLaunchKernelGGL(model_fwd_graph_kernel, dim3(B), dim3(256), 0, s, a);
    return check_launch("model_fwd_graph");
  }
  const int NP = a.N <= 32 ? 32 : (a.N <= 64 ? 64 : 128);
  const size_t lds = fwd64_lds_layout(NP, a.H, a.Z).bytes;
  GVAE_REQUIRE(lds <= kMaxDynLds, GVAE_ELIMIT, "model_fwd: %zu bytes of LDS", lds);
  auto launch = [&](auto kernel) -> int {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "model_fwd: cannot reserve %zu bytes of LDS: %s", lds,
                   hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(B), dim3(NP * 8), lds, s, a);
    return check_launch("model_fwd_lds");
  };
  return NP == 32   ? launch(model_fwd_lds_kernel<32>)
         : NP == 64 ? launch(model_fwd_lds_kernel<64>)
                    : launch(model_fwd_lds_kernel<128>);
}

// ---------------------------------------------------------------------------
// Backward of the same chain, from the gradients of the two hidden decoder layers down to the
// gradient of the first layer's linear output, again with a work-group per graph and the graph's
// tensors in LDS.  What is summed over the batch (weight and bias gradients, the BatchNorm
// parameters' per-graph partial sums) is left to grouped contractions / column sums afterwards.
// ---------------------------------------------------------------------------
__host__ __device__ inline FwdLds bwd_lds_layout(int NP, int H, int Z) {
  FwdLds L = fwd_lds_layout(NP, H, Z);
  L.bytes = (size_t)(L.o_vec + 4 * H + 4 * Z) * sizeof(float);   // dhe, dhn, 2 partial rows; dz parts, dmu, dlv
  return L;
}

// C tile = A (LDS) * W, W (k, n) row-major in global memory with row stride ldw, K <= 128
__device__ __forceinline__ f32x16 tile_lds_wn(const float *A, int as, int m0, const float *__restrict__ W,
                                              int ldw, int n0, int K) {
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const float *ap = A + (m0 + r) * as + 4 * half;
  const float *bp = W + (4 * half) * ldw + n0 + r;       // (a weight matrix: far below 2^31 floats)
  float4 bv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (8 * u < K) {
      bv[u].x = bp[(8 * u + 0) * ldw]; bv[u].y = bp[(8 * u + 1) * ldw];
      bv[u].z = bp[(8 * u + 2) * ldw]; bv[u].w = bp[(8 * u + 3) * ldw];
    }
  }
  f32x16 acc = {0};
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (8 * u < K) {
      const float4 av = *reinterpret_cast<const float4 *>(ap + 8 * u);
      MFMA4(av, bv[u]);
    }
  }
  return acc;
}

// The arithmetic of bn_relu_bwd_kernel for one graph.  The incoming gradient is either a buffer
// in LDS (dsrc) or the pooled vector's gradient spread over the live rows (dgv[c] * inv_r).
template <int NT, int NP>
__device__ __forceinline__ void bn_relu_bwd_lds(
    const float *dsrc, const float *dgv, float inv_r, const float *__restrict__ Yg,
    const float *__restrict__ Og, const float *__restrict__ bias, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ invstd, int nx, int N, int H,
    bool batch_stats, float *dst, int hs, float *__restrict__ dYg, float *__restrict__ dgamma_p,
    float *__restrict__ dbeta_p, float *__restrict__ dbias_p, double *shd) {
  constexpr int RP = NP / 4;
  const int tid = threadIdx.x, cl = tid & 63, rl = (tid >> 6) & 3, grp = tid >> 8;
  double(*sh)[64] = reinterpret_cast<double(*)[64]>(shd + grp * 256);
  for (int s0 = 0; s0 * 64 < H; s0 += NT / 256) {
    const int c = (s0 + grp) * 64 + cl;
    const bool cv = c < H;
    const float is = cv ? invstd[c] : 0.0f;
    // a lane's rows stay in registers between the two passes while they fit (16 rows at most);
    // the 32 rows of the largest size are read again (dsrc may be dst: a lane reads an element
    // before it writes it)
    constexpr int RC = RP <= 16 ? RP : 1;
    float xc[RC], dp[RC];
    auto load = [&](int i, float &x, float &d) {
      x = 0.0f;
      d = 0.0f;
      if (cv && i < nx) {
        const int e = i * H + c;
        x = Yg[e];                                  // centred by the forward
        const float din = dsrc ? dsrc[i * hs + c] : dgv[c] * inv_r;
        d = Og[e] > 0.0f ? din : 0.0f;
      }
    };
    double sb = 0.0, dot = 0.0;
    if constexpr (RP <= 16) {
#pragma unroll
      for (int q = 0; q < RP; ++q) load(rl + 4 * q, xc[q], dp[q]);
#pragma unroll
      for (int q = 0; q < RP; ++q)
        if (rl + 4 * q < nx) {
          sb += (double)dp[q];
          dot += (double)dp[q] * (double)xc[q];
        }
    } else {
      for (int i = rl; i < nx; i += 4) {
        load(i, xc[0], dp[0]);
        sb += (double)dp[0];
        dot += (double)dp[0] * (double)xc[0];
      }
    }
    sb = quad_rows_sum(sb, sh, rl, cl);
    dot = quad_rows_sum(dot, sh, rl, cl);
    const float ga = cv ? gamma[c] : 0.0f;
    const float dmean = batch_stats ? (float)(sb / (double)nx) : 0.0f;
    const float k = batch_stats ? (float)(dot * (double)is * (double)is / (double)nx) : 0.0f;
    double sdy = 0.0;
    auto emit = [&](int i, float x, float d) {
      float v = 0.0f;
      if (i < nx) v = (d - dmean - x * k) * is * ga;
      if (cv) {
        dst[i * hs + c] = v;
        if (dYg && i < N) dYg[(int64_t)i * H + c] = v;
      }
      sdy += (double)v;
    };
    if constexpr (RP <= 16) {
#pragma unroll
      for (int q = 0; q < RP; ++q) emit(rl + 4 * q, xc[q], dp[q]);
    } else {
#pragma unroll 2
      for (int i = rl; i < NP; i += 4) {
        load(i, xc[0], dp[0]);
        emit(i, xc[0], dp[0]);
      }
    }
    sdy = quad_rows_sum(sdy, sh, rl, cl);
    if (rl == 0 && cv) {
      dgamma_p[c] = (float)(dot * (double)is);
      dbeta_p[c] = (float)sb;
      dbias_p[c] = (float)sdy;
    }
  }
}

template <int NP>
__global__ __launch_bounds__(NP * 8, 4) void model_bwd_lds_kernel(ModelBwdArgs a) {
  constexpr int NT = NP * 8;
  extern __shared__ __align__(16) float lds[];
  const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const int N = a.N, H = a.H, Z = a.Z;
  const FwdLds L = bwd_lds_layout(NP, H, Z);
  const int AS = L.as, HS = L.hs;
  double *shd = reinterpret_cast<double *>(lds);
  float *Ah = lds + L.o_ah, *b = lds + L.o_b;
  float *dhe_s = lds + L.o_vec, *dhn_s = dhe_s + H, *part = dhn_s + H;   // part[2H]
  float *zpart = part + 2 * H, *dmu_s = zpart + 2 * Z, *dlv_s = dmu_s + Z;
  const int nx = a.n_x[g], R = a.R[g];
  const int64_t NH = (int64_t)N * H, gH = (int64_t)g * H, gZ = (int64_t)g * Z;
  const bool batch_stats = a.bn_mode[0] != 0;
  {
    const float *Ag = a.Ahat + (int64_t)g * N * N;
    for (int e = tid; e < NP * NP; e += NT) {
      const int i = e / NP, j = e - i * NP;
      Ah[i * AS + j] = (i < N && j < N) ? Ag[(int64_t)i * N + j] : 0.0f;
    }
    for (int t = tid; t < 2 * H; t += NT) dhe_s[t] = (t < H) ? a.dhe[gH + t] : a.dhn[gH + t - H];
  }
  __syncthreads();
  // dz = dhe * W_de1 + dhn * W_dn1 (two chains side by side), then the reparameterisation + KL
  for (int t = tid; t < 2 * Z; t += NT) {
    const bool nh = t >= Z;
    const int k = nh ? t - Z : t;
    const float *src = nh ? dhn_s : dhe_s;
    const float *__restrict__ W = (nh ? a.Wdn1 : a.Wde1) + k;
    float acc = 0.0f;
    for (int h = 0; h < H; ++h) acc = __fmaf_rn(src[h], W[(int64_t)h * Z], acc);
    zpart[t] = acc;
  }
  __syncthreads();
  for (int k = tid; k < Z; k += NT) {
    float ck = a.ck;
    if (a.d_up) ck *= a.d_up[1];
    const float dzv = zpart[Z + k] + zpart[k];
    const float raw = a.lvraw[gZ + k];
    const float lv = fminf(fmaxf(raw, -4.0f), 4.0f);
    const float dm = dzv + ck * a.mu[gZ + k] / (float)Z;
    float dl = dzv * a.eps[gZ + k] * 0.5f * expf(0.5f * lv) - ck * 0.5f * (1.0f - expf(lv)) / (float)Z;
    dl = (raw >= -4.0f && raw <= 4.0f) ? dl : 0.0f;
    dmu_s[k] = dm;
    dlv_s[k] = dl;
    a.dz[gZ + k] = dzv;
    a.dmu[gZ + k] = dm;
    a.dlv[gZ + k] = dl;
  }
  __syncthreads();
  // dgvec = dmu * W_mu + dlv * W_lv
  for (int t = tid; t < 2 * H; t += NT) {
    const bool lvh = t >= H;
    const int h = lvh ? t - H : t;
    const float *src = lvh ? dlv_s : dmu_s;
    const float *__restrict__ W = (lvh ? a.Wlv : a.Wmu) + h;
    float acc = 0.0f;
    for (int k = 0; k < Z; ++k) acc = __fmaf_rn(src[k], W[(int64_t)k * H], acc);
    part[t] = acc;
  }
  __syncthreads();
  float *dgv = dhe_s;   // (the hidden layers' gradients are used up)
  for (int h = tid; h < H; h += NT) {
    const float v = part[H + h] + part[h];
    dgv[h] = v;
    a.dgvec[gH + h] = v;
  }
  __syncthreads();
  const int tn = H / 32, tiles = (NP / 32) * tn;
  const bool has_tile = wave < tiles;
  const int m0 = (wave / tn) * 32, n0 = (wave % tn) * 32;
  f32x16 acc;
  // layer 2: pooling and BatchNorm + ReLU backward;  dH0 = Ahat * dY (Ahat is symmetric)
  bn_relu_bwd_lds<NT, NP>(nullptr, dgv, 1.0f / (float)R, a.Y2 + g * NH, a.H2 + g * NH, a.b2, a.g2,
                          a.mean2 + gH, a.invstd2 + gH, nx, N, H, batch_stats, b, HS, nullptr,
                          a.dgam2 + gH, a.dbet2 + gH, a.dbias2 + gH, shd);
  __syncthreads();
  if (has_tile) acc = tile_lds_lds(Ah, AS, m0, b, HS, n0, NP);
  __syncthreads();
  if (has_tile) tile_store(acc, b, HS, m0, n0, a.dH0_2 + g * NH, N, H);
  __syncthreads();
  // layer 1: dH1 = dH0 * W2;  BatchNorm + ReLU backward;  dH0 = Ahat * dY
  if (has_tile) acc = tile_lds_wn(b, HS, m0, a.W2, H, n0, H);
  __syncthreads();
  if (has_tile) {
    const int lane = tid & 63, col = n0 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 16; ++q)
      b[(m0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * HS + col] = acc[q];
  }
  __syncthreads();
  bn_relu_bwd_lds<NT, NP>(b, nullptr, 0.0f, a.Y1 + g * NH, a.H1 + g * NH, a.b1, a.g1, a.mean1 + gH,
                          a.invstd1 + gH, nx, N, H, batch_stats, b, HS, a.dY + g * NH, a.dgam1 + gH,
                          a.dbet1 + gH, a.dbias1 + gH, shd);
  __syncthreads();
  if (has_tile) {
    acc = tile_lds_lds(Ah, AS, m0, b, HS, n0, NP);
    // (the last result goes to the workspace only)
    float *glob = a.dH0_1 + g * NH;
    const int lane = tid & 63, col = n0 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = m0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      if (row < N) glob[(int64_t)row * H + col] = acc[q];
    }
  }
}

bool model_chain_resident(int N, int H, int Z) {
  return N <= 128 && H <= 128 && H % 32 == 0 && Z % 4 == 0 && Z <= 128;
}

int launch_model_bwd_graph(const ModelBwdArgs &a, int B, hipStream_t s) {
  GVAE_REQUIRE(model_chain_resident(a.N, a.H, a.Z), GVAE_ELIMIT,
               "model_bwd: sizes outside the LDS-resident kernel (N=%d H=%d Z=%d)", a.N, a.H, a.Z);
  const int NP = a.N <= 32 ? 32 : (a.N <= 64 ? 64 : 128);
  const size_t lds = bwd_lds_layout(NP, a.H, a.Z).bytes;
  GVAE_REQUIRE(lds <= kMaxDynLds, GVAE_ELIMIT, "model_bwd: %zu bytes of LDS", lds);
  auto launch = [&](auto kernel) -> int {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "model_bwd: cannot reserve %zu bytes of LDS: %s", lds,
                   hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(B), dim3(NP * 8), lds, s, a);
    return check_launch("model_bwd_lds");
  };
  return NP == 32   ? launch(model_bwd_lds_kernel<32>)
         : NP == 64 ? launch(model_bwd_lds_kernel<64>)
                    : launch(model_bwd_lds_kernel<128>);
}

// dmu = dz + ck * mu / Z ;  dlv = dz * eps * 0.5 * exp(0.5 lv) - ck * 0.5 * (1 - exp(lv)) / Z,
// zero where lvraw was clamped;  ck = c_kl * d_up[1] / B  (KL term, model.py:361,369)
__global__ void reparam_bwd_kernel(const float *__restrict__ dz, const float *__restrict__ mu,
                                   const float *__restrict__ lvraw, const float *__restrict__ eps,
                                   int n, int Z, float ck, const float *__restrict__ d_up,
                                   float *__restrict__ dmu, float *__restrict__ dlvraw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (d_up) ck *= d_up[1];
  const float raw = lvraw[e];
  const float lv = fminf(fmaxf(raw, -4.0f), 4.0f);
  const float g = dz[e];
  dmu[e] = g + ck * mu[e] / (float)Z;
  const float dlv = g * eps[e] * 0.5f * expf(0.5f * lv) - ck * 0.5f * (1.0f - expf(lv)) / (float)Z;
  dlvraw[e] = (raw >= -4.0f && raw <= 4.0f) ? dlv : 0.0f;
}

int launch_reparam_bwd(const float *dz, const float *mu, const float *lvraw, const float *eps, int B,
                       int Z, float c_kl, const float *d_up, float *dmu, float *dlvraw,
                       hipStream_t s) {
  const int n = B * Z;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, dz, mu, lvraw, eps,
                     n, Z, c_kl / (float)B, d_up, dmu, dlvraw);
  return check_launch("reparam_bwd");
}

// index of pair (a,b), a < b, in row-major upper-triangle order (torch.triu_indices(N,N,1))
__device__ __forceinline__ int tri_index(int a, int b, int N) {
  return a * N - (a * (a + 1)) / 2 + (b - a - 1);
}

// P = sigmoid(dense logits): off-diagonal from the edge head (symmetric), diagonal
// from the node head (reference model.py:305-307,315)
__global__ void prob_kernel(const float *__restrict__ vec, const float *__restrict__ node, int N,
                            int U, float *__restrict__ P) {
  const int g = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += gridDim.x * blockDim.x) {
    const int a = e / N, b = e - a * N;
    const float x = (a == b) ? node[(int64_t)g * N + a]
                             : vec[(int64_t)g * U + (a < b ? tri_index(a, b, N) : tri_index(b, a, N))];
    P[(int64_t)g * N * N + e] = 1.0f / (1.0f + expf(-x));
  }
}

int launch_prob(const float *vec, const float *node, int B, int N, float *P, hipStream_t s) {
  const int U = N * (N - 1) / 2;
  hipLaunchKernelGGL(prob_kernel, dim3(min(ceil_div(N * N, 256), 64), B), dim3(256), 0, s, vec, node,
                     N, U, P);
  return check_launch("prob");
}

// ---------------------------------------------------------------------------
// Losses, one work-group per graph (reference model.py:325-361):
//   edge BCE-with-logits (pos_weight) over the R(R-1)/2 permuted pairs, mean
//   node BCE-with-logits over N entries, targets 1 for k < R, weight clamp((N-R)/(R+1e-8),1,5)
//   KL = -0.5 * mean_Z(1 + lv - mu^2 - exp(lv))
// mask_dummy != 0 is the older loss (model.py:403-408,425-427): node logits of padded
// entries are read as the constant -10, in the loss and hence (zero) in its gradient.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bce_logits(float x, float y, float pw) {
  const float lw = 1.0f + (pw - 1.0f) * y;
  return (1.0f - y) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f));
}

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}
__device__ __forceinline__ float block_max_256(float v, float *sh) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(
    const float *__restrict__ vec, const float *__restrict__ node, const float *__restrict__ adj,
    const float *__restrict__ mu, const float *__restrict__ lvraw, const int64_t *__restrict__ perm,
    const int32_t *__restrict__ Rarr, int N, int Z, const float *__restrict__ pos_weight,
    int mask_dummy, float *__restrict__ rec_i, float *__restrict__ kl_i,
    float *__restrict__ maxlogit_i) {
  __shared__ float sh[4];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int R = Rarr[g], U = N * (N - 1) / 2;
  const float pw = pos_weight[0];
  const float *vg = vec + (int64_t)g * U, *ng = node + (int64_t)g * N;
  const float *Ag = adj + (int64_t)g * N * N;
  const int64_t *pg = perm + (int64_t)g * N;
  // edge term: pair t = (r,c), r < c < R in row-major order
  const int npairs = R * (R - 1) / 2;
  float s = 0.0f;
  for (int t = tid; t < npairs; t += 256) {
    // invert t -> (r,c): r is the largest row with start(r) <= t, start(r) = r*R - r(r+1)/2
    int r = (int)((2.0f * R - 1.0f - sqrtf((2.0f * R - 1.0f) * (2.0f * R - 1.0f) - 8.0f * t)) * 0.5f);
    while (r > 0 && r * R - (r * (r + 1)) / 2 > t) --r;
    while ((r + 1) * R - ((r + 1) * (r + 2)) / 2 <= t) ++r;
    const int c = t - (r * R - (r * (r + 1)) / 2) + r + 1;
    const int pr = (int)pg[r], pc = (int)pg[c];
    const float x = vg[pr < pc ? tri_index(pr, pc, N) : tri_index(pc, pr, N)];
    const float y = Ag[(int64_t)pr * N + pc];
    s += bce_logits(x, y, pw);
  }
  const float edge = block_sum_256(s, sh) / (float)npairs;
  // node term
  const float pwn = fminf(fmaxf((float)(N - R) / ((float)R + 1e-8f), 1.0f), 5.0f);
  s = 0.0f;
  for (int k = tid; k < N; k += 256)
    s += bce_logits(mask_dummy && k >= R ? -10.0f : ng[k], k < R ? 1.0f : 0.0f, pwn);
  const float nodel = block_sum_256(s, sh) / (float)N;
  // KL
  s = 0.0f;
  for (int k = tid; k < Z; k += 256) {
    const float m = mu[(int64_t)g * Z + k];
    const float lv = fminf(fmaxf(lvraw[(int64_t)g * Z + k], -4.0f), 4.0f);
    s += 1.0f + lv - m * m - expf(lv);
  }
  const float kl = -0.5f * (block_sum_256(s, sh) / (float)Z);
  // diagnostics: max |logit| over the edge head and the node head with padded
  // entries read as -10 (model.py:347-356)
  float mx = 0.0f;
  for (int t = tid; t < U; t += 256) mx = fmaxf(mx, fabsf(vg[t]));
  for (int k = tid; k < N; k += 256) mx = fmaxf(mx, k < R ? fabsf(ng[k]) : 10.0f);
  mx = block_max_256(mx, sh);
  if (tid == 0) {
    rec_i[g] = edge + nodel;
    kl_i[g] = kl;
    maxlogit_i[g] = mx;
  }
}

// out4 = [mean rec_i, mean kl_i, max maxlogit_i, max_k clamp(lvraw[B-1,k])]
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float *__restrict__ rec_i,
                                                          const float *__restrict__ kl_i,
                                                          const float *__restrict__ maxlogit_i,
                                                          const float *__restrict__ lvraw, int B,
                                                          int Z, float *__restrict__ out4) {
  __shared__ float sh[4];
  const int tid = threadIdx.x;
  float a = 0.0f, b = 0.0f, m = -INFINITY, l = -INFINITY;
  for (int g = tid; g < B; g += 256) {
    a += rec_i[g];
    b += kl_i[g];
    m = fmaxf(m, maxlogit_i[g]);
  }
  for (int k = tid; k < Z; k += 256)
    l = fmaxf(l, fminf(fmaxf(lvraw[(int64_t)(B - 1) * Z + k], -4.0f), 4.0f));
  a = block_sum_256(a, sh);
  b = block_sum_256(b, sh);
  m = block_max_256(m, sh);
  l = block_max_256(l, sh);
  if (tid == 0) {
    out4[0] = a / (float)B;
    out4[1] = b / (float)B;
    out4[2] = m;
    out4[3] = l;
  }
}

int launch_loss_fwd(const float *vec, const float *node, const float *adj, const float *mu,
                    const float *lvraw, const int64_t *perm, const int32_t *R, int B, int N, int Z,
                    const float *pos_weight, int mask_dummy, float *rec_i, float *kl_i,
                    float *maxlogit_i, float *out4, hipStream_t s) {
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(B), dim3(256), 0, s, vec, node, adj, mu, lvraw, perm, R, N,
                     Z, pos_weight, mask_dummy, rec_i, kl_i, maxlogit_i);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, rec_i, kl_i, maxlogit_i, lvraw, B,
                     Z, out4);
  return check_launch("loss_fwd");
}

// The older per-graph-density loss (reference model_legacy.py:312-352): pairs r < c < R of the
// PERMUTED ground truth against the prediction in place; pos_weight = clamp((1 - rho) /
// (rho + 1e-8), 1, 20) with rho = positives / (R(R-1)/2 + 1e-8) of this graph; no node term.
// The weight is kept per graph for the backward.
__global__ __launch_bounds__(256) void loss_fwd_indiv_kernel(
    const float *__restrict__ vec, const float *__restrict__ adj, const float *__restrict__ mu,
    const float *__restrict__ lvraw, const int64_t *__restrict__ perm,
    const int32_t *__restrict__ Rarr, int N, int Z, float *__restrict__ rec_i,
    float *__restrict__ kl_i, float *__restrict__ maxlogit_i, float *__restrict__ pw_i) {
  __shared__ float sh[4];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int R = Rarr[g], U = N * (N - 1) / 2;
  const float *vg = vec + (int64_t)g * U;
  const float *Ag = adj + (int64_t)g * N * N;
  const int64_t *pg = perm + (int64_t)g * N;
  const int npairs = R * (R - 1) / 2;
  auto pair_of = [&](int t, int &r, int &c) {
    r = (int)((2.0f * R - 1.0f - sqrtf((2.0f * R - 1.0f) * (2.0f * R - 1.0f) - 8.0f * t)) * 0.5f);
    while (r > 0 && r * R - (r * (r + 1)) / 2 > t) --r;
    while ((r + 1) * R - ((r + 1) * (r + 2)) / 2 <= t) ++r;
    c = t - (r * R - (r * (r + 1)) / 2) + r + 1;
  };
  float s = 0.0f;
  for (int t = tid; t < npairs; t += 256) {
    int r, c;
    pair_of(t, r, c);
    s += Ag[pg[r] * N + pg[c]];
  }
  const float pos = block_sum_256(s, sh);                  // a count: exact
  const float rho = pos / (float)((double)npairs + 1e-8);
  const float pw = fminf(fmaxf((1.0f - rho) / (rho + 1e-8f), 1.0f), 20.0f);
  s = 0.0f;
  for (int t = tid; t < npairs; t += 256) {
    int r, c;
    pair_of(t, r, c);
    s += bce_logits(vg[tri_index(r, c, N)], Ag[pg[r] * N + pg[c]], pw);
  }
  const float edge = block_sum_256(s, sh) / (float)npairs;
  s = 0.0f;
  for (int k = tid; k < Z; k += 256) {
    const float m = mu[(int64_t)g * Z + k];
    const float lv = fminf(fmaxf(lvraw[(int64_t)g * Z + k], -4.0f), 4.0f);
    s += 1.0f + lv - m * m - expf(lv);
  }
  const float kl = -0.5f * (block_sum_256(s, sh) / (float)Z);
  float mx = 0.0f;
  for (int t = tid; t < U; t += 256) mx = fmaxf(mx, fabsf(vg[t]));
  mx = block_max_256(mx, sh);
  if (tid == 0) {
    rec_i[g] = edge;
    kl_i[g] = kl;
    maxlogit_i[g] = mx;
    pw_i[g] = pw;
  }
}

int launch_loss_fwd_indiv(const float *vec, const float *adj, const float *mu, const float *lvraw,
                          const int64_t *perm, const int32_t *R, int B, int N, int Z, float *rec_i,
                          float *kl_i, float *maxlogit_i, float *pw_i, float *out4, hipStream_t s) {
  hipLaunchKernelGGL(loss_fwd_indiv_kernel, dim3(B), dim3(256), 0, s, vec, adj, mu, lvraw, perm, R, N,
                     Z, rec_i, kl_i, maxlogit_i, pw_i);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, rec_i, kl_i, maxlogit_i, lvraw, B,
                     Z, out4);
  return check_launch("loss_fwd_indiv");
}

// its gradient: dvec[g,(a,b)] = cr/npairs * ((1-y) - lw*(1-sigmoid(x))), y = A[perm a, perm b],
// for a < b < R, else 0; the node head gets none
__global__ void loss_bwd_indiv_kernel(const float *__restrict__ vec, const float *__restrict__ adj,
                                      const int32_t *__restrict__ Rarr,
                                      const int64_t *__restrict__ perm,
                                      const float *__restrict__ pw_i, int N, float cr,
                                      const float *__restrict__ d_up, float *__restrict__ dvec,
                                      float *__restrict__ dnode) {
  const int g = blockIdx.y;
  const int R = Rarr[g], U = N * (N - 1) / 2;
  if (d_up) cr *= d_up[0];
  const float pw = pw_i[g];
  const float inv_pairs = 1.0f / (float)(R * (R - 1) / 2);
  const float *Ag = adj + (int64_t)g * N * N;
  const int64_t *pg = perm + (int64_t)g * N;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += gridDim.x * blockDim.x) {
    const int a = e / N, b = e - a * N;
    if (a < b) {
      const int t = tri_index(a, b, N);
      float gr = 0.0f;
      if (b < R) {
        const float x = vec[(int64_t)g * U + t], y = Ag[pg[a] * N + pg[b]];
        const float lw = 1.0f + (pw - 1.0f) * y;
        const float sg = 1.0f / (1.0f + expf(-x));
        gr = cr * inv_pairs * ((1.0f - y) - lw * (1.0f - sg));
      }
      dvec[(int64_t)g * U + t] = gr;
    } else if (a == b) {
      dnode[(int64_t)g * N + a] = 0.0f;
    }
  }
}

int launch_loss_bwd_indiv(const float *vec, const float *adj, const int32_t *R, const int64_t *perm,
                          const float *pw_i, int B, int N, float c_rec, const float *d_up, float *dvec,
                          float *dnode, hipStream_t s) {
  hipLaunchKernelGGL(loss_bwd_indiv_kernel, dim3(min(ceil_div(N * N, 256), 64), B), dim3(256), 0, s,
                     vec, adj, R, perm, pw_i, N, c_rec / (float)B, d_up, dvec, dnode);
  return check_launch("loss_bwd_indiv");
}

// d(rec)/d(logits).  rec = mean_g (edge_g + node_g); the edge mean runs over all
// unordered pairs of real nodes, each exactly once whatever the permutation, so
//   dvec[g,(a,b)] = cr/npairs * ((1-y) - lw*(1-sigmoid(x)))   for a < b < R, else 0
// with cr = c_rec * d_up[0] / B.
__global__ void loss_bwd_kernel(const float *__restrict__ vec, const float *__restrict__ node,
                                const float *__restrict__ adj, const int32_t *__restrict__ Rarr,
                                int N, const float *__restrict__ pos_weight, int mask_dummy,
                                float cr, const float *__restrict__ d_up, float *__restrict__ dvec,
                                float *__restrict__ dnode) {
  const int g = blockIdx.y;
  const int R = Rarr[g], U = N * (N - 1) / 2;
  if (d_up) cr *= d_up[0];
  const float pw = pos_weight[0];
  const float inv_pairs = 1.0f / (float)(R * (R - 1) / 2);
  const float *Ag = adj + (int64_t)g * N * N;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += gridDim.x * blockDim.x) {
    const int a = e / N, b = e - a * N;
    if (a < b) {
      const int t = tri_index(a, b, N);
      float gr = 0.0f;
      if (b < R) {
        const float x = vec[(int64_t)g * U + t], y = Ag[e];
        const float lw = 1.0f + (pw - 1.0f) * y;
        const float sg = 1.0f / (1.0f + expf(-x));
        gr = cr * inv_pairs * ((1.0f - y) - lw * (1.0f - sg));
      }
      dvec[(int64_t)g * U + t] = gr;
    } else if (a == b) {
      const float pwn = fminf(fmaxf((float)(N - R) / ((float)R + 1e-8f), 1.0f), 5.0f);
      const float x = node[(int64_t)g * N + a], y = a < R ? 1.0f : 0.0f;
      const float lw = 1.0f + (pwn - 1.0f) * y;
      const float sg = 1.0f / (1.0f + expf(-x));
      dnode[(int64_t)g * N + a] =
          mask_dummy && a >= R ? 0.0f : cr / (float)N * ((1.0f - y) - lw * (1.0f - sg));
    }
  }
}

int launch_loss_bwd(const float *vec, const float *node, const float *adj, const int32_t *R, int B,
                    int N, const float *pos_weight, int mask_dummy, float c_rec, const float *d_up,
                    float *dvec, float *dnode, hipStream_t s) {
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(min(ceil_div(N * N, 256), 64), B), dim3(256), 0, s, vec,
                     node, adj, R, N, pos_weight, mask_dummy, c_rec / (float)B, d_up, dvec, dnode);
  return check_launch("loss_bwd");
}

}  // namespace gvae
