// Shared host-side plumbing for libfnssl_hip.so: error reporting across the C
// ABI (no exceptions), launch checking and optional per-kernel HIP-event timing.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/fnssl.h"

namespace fnssl {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// Per-kernel timing (enabled by fnssl_timing_enable): records a hipEvent pair
// around the launch on the launch stream; collected by fnssl_timing_collect.
struct TimedLaunch {
  TimedLaunch(const char* name, hipStream_t s, double flops = 0.0);
  ~TimedLaunch();
  const char* name_;
  hipStream_t s_;
  double flops_;
  hipEvent_t e0_ = nullptr;
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the current device (256 on MI355X); the launch planners size their rounds with it.
int device_cus();

}  // namespace fnssl

#define FNSSL_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      fnssl::set_error(__VA_ARGS__);        \
      return FNSSL_E_INVALID;               \
    }                                       \
  } while (0)

#define FNSSL_HIP(expr)                                                              \
  do {                                                                               \
    hipError_t e__ = (expr);                                                         \
    if (e__ != hipSuccess) {                                                         \
      fnssl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),       \
                       __FILE__, __LINE__);                                          \
      return FNSSL_E_HIP;                                                            \
    }                                                                                \
  } while (0)

#define FNSSL_CHECK_LAUNCH(what)                                                     \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      fnssl::set_error("launch of %s failed: %s", what, hipGetErrorString(e__));     \
      return FNSSL_E_HIP;                                                            \
    }                                                                                \
  } while (0)
