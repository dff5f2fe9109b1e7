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

}  // namespace gvae

#define GVAE_REQUIRE(cond, code, ...)                       \
  do {                                                      \
    if (!(cond)) return gvae::set_error((code), __VA_ARGS__); \
  } while (0)
