// Batched linear-sum-assignment on the GPU: one wavefront (64 lanes) per graph (the solver
// itself is in gvae_lsap.h).
#include "gvae_lsap.h"

namespace gvae {

// LDS: float cost[n*n] (= X)
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ X, int ldx,
                                                  int64_t strideX,
                                                  const int32_t *__restrict__ Rarr, int r_max,
                                                  int64_t *__restrict__ perm, int n_perm,
                                                  int32_t *__restrict__ status) {
  extern __shared__ float cost[];
  const int g = blockIdx.x, lane = threadIdx.x;
  const int n = Rarr[g];
  int64_t *pg = perm + (int64_t)g * n_perm;
  if (n <= 0) {
    for (int k = lane; k < n_perm; k += 64) pg[k] = k;
    if (lane == 0) status[g] = 0;
    return;
  }
  const float *Xg = X + g * strideX;
  for (int e = lane; e < n * n; e += 64) {
    const int r = e / n, c = e - r * n;
    cost[e] = Xg[(int64_t)r * ldx + c];
  }
  __syncthreads();
  lsap_wave(cost, n, n, lane, pg, n_perm, status + g);
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
  const size_t lds = (size_t)r_max * r_max * 4;
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
