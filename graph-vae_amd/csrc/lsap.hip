// Batched linear-sum-assignment on the GPU: one wavefront (64 lanes) per graph.
//
// Reproduces scipy.optimize.linear_sum_assignment (SciPy 1.15.3, shortest
// augmenting path) on cost = -X in double precision, including its scan order and
// tie-breaking, so the permutation is identical to the reference's host round trip
// (graphvae/models/model.py:319-323):
//   - rows are augmented in order 0..n-1;
//   - the list of unscanned columns starts as n-1, n-2, ..., 0 and shrinks by
//     moving its last entry into the freed slot;
//   - a column's tentative distance is lowered only on strict '<';
//   - the next column is, among those at the minimum distance, the last one in list
//     order that is unassigned if there is one, otherwise the first one in list order.
// The column scan (the O(n) inner loop) is spread over the 64 lanes; the selection
// rule above is evaluated with wave-wide min/max reductions, which is equivalent to
// the sequential scan.  Dual variables, distances and the path live in LDS together
// with the fp32 matrix itself.
#include "gvae_common.h"

namespace gvae {

// Wave-wide reductions on the data-parallel-primitive path (no LDS traffic): an xor butterfly
// inside each row of 16 lanes (quad permutes, then half-row and row mirrors), then the four
// row results are combined through scalar lane reads.  Every lane ends up with the result.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double min_f64(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double read_lane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_min_f64(double v) {
  v = min_f64(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
  v = min_f64(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
  v = min_f64(v, dpp_f64<0x141>(v));  // row_half_mirror
  v = min_f64(v, dpp_f64<0x140>(v));  // row_mirror
  return min_f64(min_f64(read_lane_f64(v, 0), read_lane_f64(v, 16)),
                 min_f64(read_lane_f64(v, 32), read_lane_f64(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v) {
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// LDS layout per graph (n = r_max rounded up to even):
//   double u[n], v[n], spc[n];  int path[n], col4row[n], row4col[n], remaining[n];
//   uchar SR[n], SC[n] (as ints for simplicity);  float cost[n*n] (= X, negated on read)
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ X, int ldx,
                                                  int64_t strideX,
                                                  const int32_t *__restrict__ Rarr, int r_max,
                                                  int64_t *__restrict__ perm, int n_perm,
                                                  int32_t *__restrict__ status) {
  extern __shared__ double lds_d[];
  const int g = blockIdx.x, lane = threadIdx.x;
  const int n = Rarr[g];
  const int np = (r_max + 1) & ~1;
  double *u = lds_d, *v = u + np, *spc = v + np;
  int *path = reinterpret_cast<int *>(spc + np);
  int *col4row = path + np, *row4col = col4row + np, *remaining = row4col + np;
  int *SR = remaining + np, *SC = SR + np;
  float *cost = reinterpret_cast<float *>(SC + np);

  int64_t *pg = perm + (int64_t)g * n_perm;
  for (int k = lane; k < n_perm; k += 64) pg[k] = k;
  if (n <= 0) {
    if (lane == 0) status[g] = 0;
    return;
  }
  const float *Xg = X + g * strideX;
  int bad = 0;
  for (int e = lane; e < n * n; e += 64) {
    const int r = e / n, c = e - r * n;
    const float x = Xg[(int64_t)r * ldx + c];
    cost[e] = x;
    bad |= (x != x) || (x == INFINITY);  // cost = -x: NaN or -inf is invalid
  }
  for (int k = lane; k < n; k += 64) {
    u[k] = 0.0; v[k] = 0.0; path[k] = -1; col4row[k] = -1; row4col[k] = -1;
  }
  bad = wave_max_i32(bad);
  if (bad) {
    if (lane == 0) status[g] = 1;
    return;
  }
  __syncthreads();

  int fail = 0;
  for (int cur = 0; cur < n && !fail; ++cur) {
    double minVal = 0.0;
    int i = cur, num_remaining = n, sink = -1;
    for (int k = lane; k < n; k += 64) {
      remaining[k] = n - k - 1;
      SR[k] = 0; SC[k] = 0;
      spc[k] = INFINITY;
    }
    __syncthreads();
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      const float *crow = cost + i * n;
      // pass over the unscanned columns, two chunks of 64 lanes (n <= 128)
      double sj[2];
      int free_col[2];
      double lowest = INFINITY;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int it = c * 64 + lane;
        sj[c] = INFINITY;
        free_col[c] = 0;
        if (it < num_remaining) {
          const int j = remaining[it];
          const double r = minVal + (double)(-crow[j]) - ui - v[j];
          double cur_s = spc[j];
          if (r < cur_s) { path[j] = i; spc[j] = r; cur_s = r; }
          sj[c] = cur_s;
          free_col[c] = row4col[j] == -1;
          lowest = cur_s < lowest ? cur_s : lowest;
        }
      }
      lowest = wave_min_f64(lowest);
      if (lowest == INFINITY) { fail = 1; break; }
      // among the columns at the minimum: the last unassigned one in list order if there is
      // one, else the first one in list order (what the sequential scan ends up with)
      int last_free = -1, first_any = 0x7fffffff;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bool at_min = (c * 64 + lane < num_remaining) && sj[c] == lowest;
        const unsigned long long bm = __ballot(at_min);
        const unsigned long long bf = __ballot(at_min && free_col[c]);
        if (bm != 0ull && first_any == 0x7fffffff) first_any = c * 64 + __ffsll((long long)bm) - 1;
        if (bf != 0ull) last_free = c * 64 + 63 - __clzll((long long)bf);
      }
      const int index = last_free >= 0 ? last_free : first_any;
      minVal = lowest;
      const int j = remaining[index];
      const int rj = row4col[j];
      if (rj == -1) sink = j; else i = rj;
      __syncthreads();  // all lanes have read remaining[]/row4col[] before the update
      if (lane == 0) {
        SC[j] = 1;
        remaining[index] = remaining[num_remaining - 1];
      }
      --num_remaining;
      __syncthreads();
    }
    if (fail) break;
    // dual updates
    for (int r = lane; r < n; r += 64) {
      if (r == cur) u[r] += minVal;
      else if (SR[r]) u[r] += minVal - spc[col4row[r]];
    }
    for (int c = lane; c < n; c += 64)
      if (SC[c]) v[c] -= minVal - spc[c];
    __syncthreads();
    // augment along the path (serial, wave-uniform)
    if (lane == 0) {
      int j = sink;
      for (;;) {
        const int r = path[j];
        row4col[j] = r;
        const int t = col4row[r];
        col4row[r] = j;
        j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  if (fail) {
    if (lane == 0) status[g] = 1;
    return;
  }
  // perm[col4row[i]] = i
  for (int r = lane; r < n; r += 64) pg[col4row[r]] = r;
  if (lane == 0) status[g] = 0;
}

}  // namespace gvae

using namespace gvae;

extern "C" int gvae_lsap(const float *d_X, int ldx, int64_t strideX, const int32_t *d_R,
                         int n_graphs, int r_max, int64_t *d_perm, int n_perm, int32_t *d_status,
                         void *stream) {
  GVAE_REQUIRE(d_X && d_R && d_perm && d_status, GVAE_EINVAL, "lsap: null pointer");
  GVAE_REQUIRE(n_graphs >= 0 && r_max > 0 && r_max <= ldx && n_perm >= r_max, GVAE_EINVAL,
               "lsap: bad sizes (n_graphs=%d r_max=%d ldx=%d n_perm=%d)", n_graphs, r_max, ldx,
               n_perm);
  const int np = (r_max + 1) & ~1;
  const size_t lds = (size_t)np * (3 * 8 + 6 * 4) + (size_t)r_max * r_max * 4;
  GVAE_REQUIRE(r_max <= GVAE_MAX_NODES_LDS && lds <= kMaxDynLds, GVAE_ELIMIT,
               "lsap: r_max=%d needs %zu bytes of LDS (limit 160 KiB)", r_max, lds);
  if (n_graphs == 0) return GVAE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)lsap_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "lsap: cannot reserve %zu bytes of LDS: %s", lds,
                 hipGetErrorString(e));
  }
  ProfileScope ps(F_LSAP, s);
  hipLaunchKernelGGL(lsap_kernel, dim3(n_graphs), dim3(64), lds, s, d_X, ldx, strideX, d_R, r_max,
                     d_perm, n_perm, d_status);
  return check_launch("lsap");
}
