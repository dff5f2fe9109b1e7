// Error reporting and the event profiler of libgvae_hip.so.
#include "gvae_common.h"

#include <mutex>
#include <vector>

namespace gvae {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(GVAE_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return GVAE_OK;
}

struct EventPair {
  int family;
  hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<EventPair> g_events;   // pairs recorded since the last reset
static std::vector<EventPair> g_pool;     // created events kept for reuse (creation is slow)

ProfileScope::ProfileScope(int fam, hipStream_t s) : family(fam), stream(s), slot(-1) {
  if (!g_prof_on || fam < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  EventPair p{fam, nullptr, nullptr};
  if (!g_pool.empty()) {
    p = g_pool.back();
    p.family = fam;
    g_pool.pop_back();
  } else if (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess) {
    return;
  }
  hipEventRecord(p.start, s);
  g_events.push_back(p);
  slot = (int)g_events.size() - 1;
}

ProfileScope::~ProfileScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(g_events[slot].stop, stream);
}

}  // namespace gvae

extern "C" {

int gvae_version(void) { return 100; }

const char *gvae_last_error(void) { return gvae::g_err; }

int gvae_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(gvae::g_prof_mu);
  gvae::g_prof_on = on != 0;
  return GVAE_OK;
}

int gvae_profile_reset(void) {
  std::lock_guard<std::mutex> lk(gvae::g_prof_mu);
  for (auto &p : gvae::g_events) gvae::g_pool.push_back(p);
  gvae::g_events.clear();
  return GVAE_OK;
}

int gvae_profile_read(int family, double *total_ms, int64_t *launches) {
  std::lock_guard<std::mutex> lk(gvae::g_prof_mu);
  double tot = 0.0;
  int64_t n = 0;
  for (auto &p : gvae::g_events) {
    if (p.family != family) continue;
    if (hipEventSynchronize(p.stop) != hipSuccess)
      return gvae::set_error(GVAE_ELAUNCH, "profile_read: event synchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.start, p.stop) != hipSuccess)
      return gvae::set_error(GVAE_ELAUNCH, "profile_read: elapsed time failed");
    tot += ms;
    ++n;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return GVAE_OK;
}

}  // extern "C"
