// Read-bandwidth of dynamically drawn chunks under different chunk -> XCD assignments.
// Every work-group draws chunk numbers from the counter of the XCD it runs on; `mode` maps
// the n-th draw of XCD k to a chunk.  build: hipcc -O3 --offload-arch=gfx950 read_xcd.hip -o read_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}

// mode 0: one shared counter (chunk = draw)            1: chunk = 8*n + k (interleaved)
// mode 2: chunk = k*(N/8) + n (contiguous eighths)     3: groups of 8 chunks: group = 8*(n/8)+k
// mode 4: hashed: chunk = perm(8*n + k) with an odd multiplier (fixed subsets, no regularity)
// mode 5: shared counter, but 8 counters round-robin by blockIdx (no XCD relation): chunk = 8*n + (blockIdx%8)... uses xcc of block 0? no: k = blockIdx.x & 7
__global__ void rd_kernel(const f32x4_t *__restrict__ S, long n_chunks, long chunk_vecs,
                          unsigned *__restrict__ counters, float *__restrict__ out, int mode) {
  __shared__ long s_c;
  const int tid = threadIdx.x, nt = blockDim.x;
  float m = 0.f;
  const int k = mode == 5 ? (blockIdx.x >> 3) & 7 : xcc_id();
  for (;;) {
    if (tid == 0) {
      long c;
      if (mode == 0) {
        c = atomicAdd(counters, 1u);
      } else {
        const long n = atomicAdd(counters + 32 * k, 1u);
        const long per = n_chunks / 8;
        if (n >= per) c = n_chunks;
        else if (mode == 1 || mode == 5) c = 8 * n + k;
        else if (mode == 2) c = k * per + n;
        else if (mode == 3) c = ((n >> 3) * 8 + k) * 8 + (n & 7);
        else c = (((8 * n + k) * 40503L) % n_chunks);
      }
      s_c = c;
    }
    __syncthreads();
    const long c = s_c;
    __syncthreads();
    if (c >= n_chunks) break;
    const f32x4_t *p = S + c * chunk_vecs;
    for (long v = tid; v < chunk_vecs; v += (long)nt * 4) {
      f32x4_t x[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) x[d] = __builtin_nontemporal_load(p + v + (long)d * nt);
#pragma unroll
      for (int d = 0; d < 4; ++d) m = fmaxf(m, fmaxf(fmaxf(x[d].x, x[d].y), fmaxf(x[d].z, x[d].w)));
    }
  }
  if (m == 12345.f) out[blockIdx.x] = m;
}

int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 8) << 30;
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
  const char *names[] = {"shared counter", "xcd: interleaved chunks", "xcd: contiguous eighths",
                         "xcd: interleaved groups of 8", "xcd: hashed subsets", "8 counters by blockIdx/8"};
  for (size_t chunk : {(size_t)128 << 10, (size_t)512 << 10, (size_t)4 << 20}) {
    for (int mode = 0; mode < 6; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipMemsetAsync(counters, 0, 4096, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rd_kernel, dim3(2048), dim3(256), 0, 0, S, (long)(bytes / chunk),
                           (long)(chunk / 16), counters, out, mode);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("chunk %5zu KiB  %-30s %8.3f ms  %6.3f TB/s\n", chunk >> 10, names[mode], best,
             bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
