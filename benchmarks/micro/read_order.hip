// Read bandwidth of 256-thread work-groups that each stream whole chunks, by the rule that
// maps work-groups to chunks.  build: hipcc -O3 --offload-arch=gfx950 read_order.hip -o read_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}
// mode 0: one shared counter              1: static, chunk = block + k*grid
// mode 2: static, scrambled (odd multiplier modulo the power-of-two chunk count)
// mode 3: shared counter, scrambled       4: per-XCD counters + stealing, chunk = 8*n + xcd
// mode 5: per-XCD counters + stealing, scrambled
__global__ void rd_kernel(const f32x4_t *__restrict__ S, long n_chunks, long chunk_vecs,
                          unsigned *__restrict__ counters, float *__restrict__ out, int mode) {
  __shared__ long s_c;
  const int tid = threadIdx.x, nt = blockDim.x;
  float m = 0.f;
  long k = 0;
  const int x = xcc_id();
  int steal = 0;
  for (;;) {
    if (tid == 0) {
      long c;
      if (mode == 0 || mode == 3) c = atomicAdd(counters, 1u);
      else if (mode == 1 || mode == 2) c = blockIdx.x + (k++) * gridDim.x;
      else {
        const long per = n_chunks / 8;
        c = n_chunks;
        for (; steal < 8; ++steal) {
          const int q = (x + steal) & 7;
          const long n = atomicAdd(counters + 32 * q, 1u);
          if (n < per) { c = 8 * n + q; break; }
        }
      }
      if ((mode == 2 || mode == 3 || mode == 5) && c < n_chunks) c = (c * 40503L) & (n_chunks - 1);
      s_c = c;
    }
    __syncthreads();
    const long c = s_c;
    __syncthreads();
    if (c >= n_chunks) break;
    const f32x4_t *p = S + c * chunk_vecs;
    for (long v = tid; v < chunk_vecs; v += (long)nt * 4) {
      f32x4_t xx[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) xx[d] = __builtin_nontemporal_load(p + v + (long)d * nt);
#pragma unroll
      for (int d = 0; d < 4; ++d) m = fmaxf(m, fmaxf(fmaxf(xx[d].x, xx[d].y), fmaxf(xx[d].z, xx[d].w)));
    }
  }
  if (m == 12345.f) out[blockIdx.x] = m;
}

int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 32) << 30;
  const int grid = argc > 2 ? atoi(argv[2]) : 1280;
  f32x4_t *S;
  unsigned *counters;
  float *out;
  if (hipMalloc(&S, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&counters, 4096);
  hipMalloc(&out, 1 << 20);
  hipMemset(S, 0, bytes);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *names[] = {"shared counter", "static strided", "static scrambled", "shared counter scrambled",
                         "xcd counters + stealing", "xcd counters + stealing scrambled"};
  for (size_t chunk : {(size_t)128 << 10, (size_t)512 << 10}) {
    for (int mode = 0; mode < 6; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(counters, 0, 4096, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rd_kernel, dim3(grid), dim3(256), 0, 0, S, (long)(bytes / chunk),
                           (long)(chunk / 16), counters, out, mode);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("grid %d chunk %5zu KiB  %-34s %8.3f ms  %6.3f TB/s\n", grid, chunk >> 10, names[mode], best,
             bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
