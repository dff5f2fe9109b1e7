// Host-side plumbing shared by the .hip translation units: error text, launch
// checks and the optional per-kernel-family event profiler.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/gvae_hip.h"

namespace gvae {

int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int check_launch(const char *what);

enum Family { F_EDGE_SIM = 0, F_MPM_ITER = 1, F_MPM_NORM = 2, F_MPM_FUSED = 3, F_LSAP = 4,
              F_NODE_SIM = 5, F_COUNT = 16 };

// RAII bracket: records a start event on construction and a stop event on
// destruction when profiling is enabled; otherwise does nothing.
struct ProfileScope {
  ProfileScope(int family, hipStream_t s);
  ~ProfileScope();
  int family;
  hipStream_t stream;
  int slot;
};

// dynamic LDS a kernel may request: 160 KiB per work-group minus room for its static LDS
constexpr size_t kMaxDynLds = 160 * 1024 - 1024;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// gvae_match_stream in parts (matching.hip): match_stream_run = match_stream_prep (similarity
// tensor, first iteration), match_stream_iterate (the persistent streaming launch) and, if
// run_tail, match_stream_tail (the S-free kernel for the graphs the launch handed over).  A caller
// may enqueue the three separately, the tail on another stream behind the launch.
int match_stream_prep(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                      int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                      const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                      float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream);
int match_stream_iterate(const int32_t *d_R, int n_graphs, int r_max, int s_layout, const float *d_S,
                         const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                         float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream);
int match_stream_run(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                     int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                     const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                     float tol, int early_stop, int32_t *d_iters_done, void *d_ws, bool run_tail,
                     void *stream);
int match_stream_tail(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                      int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                      int64_t strideX, int iters, float tol, int early_stop, int32_t *d_iters_done,
                      void *d_ws, void *stream);

}  // namespace gvae

#define GVAE_REQUIRE(cond, code, ...)                       \
  do {                                                      \
    if (!(cond)) return gvae::set_error((code), __VA_ARGS__); \
  } while (0)
