// Read-bandwidth ceiling for the access pattern of the streamed matching: each work-group
// reads contiguous chunks (one work item = `chunk` bytes) in passes of NT*16*DEPTH bytes and
// keeps DEPTH 16-byte loads per lane in flight.  Reports TB/s for a few shapes.
// build: hipcc -O3 --offload-arch=gfx950 read_stream.hip -o read_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int DEPTH, bool NT_LOAD>
__global__ void rd_kernel(const f32x4_t *__restrict__ S, long n_chunks, long chunk_vecs,
                          unsigned *__restrict__ counter, float *__restrict__ out, int dynamic) {
  __shared__ long s_c;
  const int tid = threadIdx.x, nt = blockDim.x;
  float m = 0.f;
  long c = blockIdx.x;
  for (;;) {
    if (dynamic) {
      if (tid == 0) s_c = atomicAdd(counter, 1u);
      __syncthreads();
      c = s_c;
      if (dynamic == 2) c = ((long)__builtin_amdgcn_readfirstlane((int)(c >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)c);
      __syncthreads();
    }
    if (c >= n_chunks) break;
    const f32x4_t *p = S + c * chunk_vecs;
    for (long v = tid; v < chunk_vecs; v += (long)nt * DEPTH) {
      f32x4_t x[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const long idx = v + (long)d * nt;
        const f32x4_t *q = p + (idx < chunk_vecs ? idx : v);
        x[d] = NT_LOAD ? __builtin_nontemporal_load(q) : *q;
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) m = fmaxf(m, fmaxf(fmaxf(x[d].x, x[d].y), fmaxf(x[d].z, x[d].w)));
    }
    if (!dynamic) {
      c += gridDim.x;
    }
  }
  if (m == 12345.f) out[blockIdx.x] = m;
}

template <int DEPTH, bool NT_LOAD>
static void run(const char *name, const f32x4_t *S, size_t bytes, size_t chunk, int nt, int grid,
                int dynamic, unsigned *counter, float *out) {
  const long n_chunks = bytes / chunk, chunk_vecs = chunk / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemsetAsync(counter, 0, 4, 0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rd_kernel<DEPTH, NT_LOAD>), dim3(grid), dim3(nt), 0, 0, S, n_chunks, chunk_vecs,
                       counter, out, dynamic);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-34s nt=%4d grid=%6d chunk=%7zu KiB depth=%d %s: %8.3f ms  %6.3f TB/s\n", name, nt, grid,
         chunk >> 10, DEPTH, dynamic == 2 ? "dynU" : dynamic ? "dyn " : "grid", best, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 8) << 30;
  f32x4_t *S;
  unsigned *counter;
  float *out;
  if (hipMalloc(&S, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&counter, 256);
  hipMalloc(&out, 1 << 20);
  hipMemset(S, 0, bytes);
  hipDeviceSynchronize();
  for (size_t chunk : {(size_t)512 << 10, (size_t)4 << 20}) {
    run<4, true>("nt256 d4 persistent dyn 8/CU", S, bytes, chunk, 256, 2048, 1, counter, out);
    run<4, true>("nt256 d4 persistent dynU 8/CU", S, bytes, chunk, 256, 2048, 2, counter, out);
    run<4, true>("nt256 d4 persistent static 8/CU", S, bytes, chunk, 256, 2048, 0, counter, out);
    run<4, true>("nt256 d4 persistent dynU 7/CU", S, bytes, chunk, 256, 1792, 2, counter, out);
    run<4, true>("nt256 d4 persistent dynU 6/CU", S, bytes, chunk, 256, 1536, 2, counter, out);
    run<4, true>("nt256 d4 persistent dynU 4/CU", S, bytes, chunk, 256, 1024, 2, counter, out);
    run<8, true>("nt256 d8 persistent dynU 4/CU", S, bytes, chunk, 256, 1024, 2, counter, out);
    run<4, true>("nt1024 d4 persistent dynU 1/CU", S, bytes, chunk, 1024, 256, 2, counter, out);
    run<4, true>("nt1024 d4 persistent dynU 2/CU", S, bytes, chunk, 1024, 512, 2, counter, out);
    run<4, true>("nt256 d4 one WG per chunk", S, bytes, chunk, 256, (int)(bytes / chunk), 0, counter, out);
  }
  return 0;
}
