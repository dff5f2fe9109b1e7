// Latency of the building blocks of a single-wavefront dependent chain on gfx950 (one wave per CU,
// nothing else running): cycles per iteration by s_memtime.  Build: hipcc -O3 --offload-arch=gfx950
// -I graph-vae_amd/csrc -I include benchmarks/micro/wave_latency.hip -o benchmarks/micro/wave_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "gvae_lsap.h"
using namespace gvae;

#define ITERS 2000
template <int T>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, const float *in, int n) {
  __shared__ float lds[64 * 64];
  const int lane = threadIdx.x;
  for (int e = lane; e < 64 * 64; e += 64) lds[e] = in[e];
  __syncthreads();
  double x = in[lane], acc = 0.0;
  int i = n & 31;
  unsigned long long mask = ~0ull;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (T == 0) {          // f64 min reduction over 32 lanes, dependent
      x = wave_min_f64<32>(x) + (double)lane;
    } else if constexpr (T == 1) {   // over 64 lanes
      x = wave_min_f64<64>(x) + (double)lane;
    } else if constexpr (T == 2) {   // cvt + 3 dependent f64 adds
      x = ((x + (double)(-(float)x)) - acc) - 1.0;
    } else if constexpr (T == 3) {   // readlane with a scalar index that depends on the previous result
      const int r = __builtin_amdgcn_readlane((int)x + lane, i);
      i = r & 31;
      x = x + 1.0;
    } else if constexpr (T == 4) {   // LDS read at a uniform row that depends on the previous value
      const float c = lds[i * 64 + lane];
      i = __builtin_amdgcn_readfirstlane((int)c) & 31;
    } else if constexpr (T == 5) {   // compare -> ballot -> popcount / ctz -> scalar index
      const unsigned long long m = __ballot(x > (double)i);
      i = (__popcll(m) + __builtin_ctzll(m | 1ull << 63)) & 31;
    } else if constexpr (T == 6) {   // one 4-cycle vector instruction, dependent
      x = x + 1.0;
    } else if constexpr (T == 7) {   // 32-bit integer max reduction over 32 lanes
      i = wave_max_i32<32>(i + lane) & 1023;
    } else if constexpr (T == 8) {   // vector compare -> scalar and with a mask -> select
      const bool a = __builtin_amdgcn_inverse_ballot_w64(mask);
      const bool u = a && x < acc;
      acc = u ? x : acc;
      x = x + 0.5;
      mask = (mask << 1) | (mask >> 63);
    } else if constexpr (T == 9) {   // scalar -> readlane -> scalar compare -> taken branch
      const int r = __builtin_amdgcn_readlane(lane * 3, i);
      if (r == 1000) break;
      i = (r + 7) & 31;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = x + acc + i + (double)mask;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double *out; long long *cyc; float *in;
  hipMalloc(&out, 64 * 8 * 8); hipMalloc(&cyc, 64); hipMalloc(&in, 64 * 64 * 4);
  float h[64 * 64];
  for (int e = 0; e < 64 * 64; ++e) h[e] = (float)((e * 7919) % 31);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const char *names[] = {"f64 min over 32 lanes (+1 add)", "f64 min over 64 lanes (+1 add)", "cvt + 3 f64 adds",
                         "readlane(scalar idx) -> scalar -> idx", "LDS read (uniform row) -> readfirstlane",
                         "v_cmp -> ballot -> popcnt/ctz", "one f64 add", "i32 max over 32 lanes",
                         "v_cmp + s_and + 2 cndmask + add", "readlane -> s_cmp -> branch"};
  auto run = [&](auto kern, int t) {
    long long best = 1LL << 60;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, cyc, in, 5);
      long long c;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    printf("%-45s %7.1f cycles per iteration (s_memtime ticks: 100 MHz? see note)\n", names[t], (double)best / ITERS);
  };
  run(k<0>, 0); run(k<1>, 1); run(k<2>, 2); run(k<3>, 3); run(k<4>, 4); run(k<5>, 5); run(k<6>, 6);
  run(k<7>, 7); run(k<8>, 8); run(k<9>, 9);
  return 0;
}
