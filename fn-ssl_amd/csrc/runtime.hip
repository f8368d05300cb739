// Error string, ABI version and the per-kernel HIP-event timing registry.
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace fnssl {

int device_cus() {
  // per device: a process may drive any of the node's GPUs (one process per GPU picks its own ordinal)
  static std::mutex mu;
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (cache[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev] = n;
  }
  return cache[dev];
}


static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
struct Rec {
  const char* name;
  double flops;
  hipEvent_t e0, e1;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
bool g_timing = false;
char g_select[64] = "";   // non-empty: only launches with this name are bracketed
}  // namespace

TimedLaunch::TimedLaunch(const char* name, hipStream_t s, double flops)
    : name_(name), s_(s), flops_(flops) {
  if (!g_timing) return;
  if (g_select[0] && std::strcmp(g_select, name) != 0) return;
  if (hipEventCreate(&e0_) != hipSuccess) {
    e0_ = nullptr;
    return;
  }
  (void)hipEventRecord(e0_, s_);
}

TimedLaunch::~TimedLaunch() {
  if (!e0_) return;
  hipEvent_t e1 = nullptr;
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0_);
    return;
  }
  (void)hipEventRecord(e1, s_);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(Rec{name_, flops_, e0_, e1});
}

}  // namespace fnssl

extern "C" {

int fnssl_abi_version(void) { return FNSSL_ABI_VERSION; }

const char* fnssl_last_error(void) { return fnssl::g_err; }

int fnssl_timing_enable(int enable) {
  std::lock_guard<std::mutex> lk(fnssl::g_mu);
  fnssl::g_timing = enable != 0;
  return FNSSL_OK;
}

int fnssl_timing_select(const char* name) {
  std::lock_guard<std::mutex> lk(fnssl::g_mu);
  std::snprintf(fnssl::g_select, sizeof(fnssl::g_select), "%s", name ? name : "");
  return FNSSL_OK;
}

int fnssl_timing_collect(int cap, char (*names)[64], double* total_ms, long long* count,
                         double* flops) {
  std::vector<fnssl::Rec> recs;
  {
    std::lock_guard<std::mutex> lk(fnssl::g_mu);
    recs.swap(fnssl::g_recs);
  }
  struct Agg {
    double ms = 0, flops = 0;
    long long n = 0;
  };
  std::map<std::string, Agg> agg;
  for (auto& r : recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      Agg& a = agg[r.name];
      a.ms += ms;
      a.flops += r.flops;
      a.n += 1;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  int i = 0;
  for (auto& kv : agg) {
    if (i >= cap) break;
    std::strncpy(names[i], kv.first.c_str(), 63);
    names[i][63] = 0;
    total_ms[i] = kv.second.ms;
    count[i] = kv.second.n;
    flops[i] = kv.second.flops;
    ++i;
  }
  return i;
}

}  // extern "C"
