// Error string, ABI version and the per-kernel HIP-event timing registry.
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "tuning.h"

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

// ---- tuning: the process default (set explicitly through the ABI, never from the environment), plus the descriptor of
// the call in progress.  Readers work on a per-thread snapshot that is refreshed when the default's version changes, so a
// call never takes a lock per knob and PyTorch's autograd worker threads see what the main thread set.
namespace {
std::mutex g_tuning_mu;
fnssl_tuning g_default_tuning = {sizeof(fnssl_tuning), {0}};
std::atomic<unsigned> g_tuning_version{1};
thread_local fnssl_tuning g_thread_tuning = {sizeof(fnssl_tuning), {0}};
thread_local unsigned g_thread_version = 0;
thread_local const fnssl_tuning* g_call_tuning = nullptr;
const char* const kTuneNames[] = {
    "LSTM_NO_STATIC",
    "NO_STATIC3",
    "NO_STATIC2",
    "NO_STATIC_IPDNET",
    "LSTM_SPLIT",
    "SPLIT4_MAX_H256",
    "LSTM_VARIANT_H128",
    "LSTM_VARIANT_H256",
    "LSTM_CHQ",
    "NO_F32_CLUSTER",
    "NO_F32C_B1",
    "TRAIN_NO_F32_CLUSTER",
    "F32C_NO_ROTATE",
    "F32C_PRIO",
    "NO_CLUSTER",
    "NO_CLUSTER_B1",
    "NO_CLUSTER_H128",
    "CLUSTER_SPREAD",
    "BF16P_DRAIN",
    "BF16W_SOLO",
    "BWD_NO_CLUSTER",
    "BWD_CLUSTER_MIN_GROUPS",
    "BWD_CLUSTER_NO_ROTATE",
    "BWDC_NO_PREFETCH",
    "BWDC_NO_TOKEN",
    "BWDC_WAVES16",
    "FWD_RING",
    "BWD_RING",
    "TRAIN_SPLIT",
    "TRAIN_NO_STATIC",
    "NO_FWD2",
    "NO_BWD2",
    "SN_SCALAR",
    "STFT_PER_FRAME",
    "CLUSTER_SPIN_LIMIT",
    "CLUSTER_TEST_STALL",
    "RESERVED_CUS",
    "NO_F32_SMALL",
    "F32C_MIN_GROUPS",
    "F32C_GATE_SPLIT",
    "CLUSTER_FULL_TILES",
    "NO_STATIC4"};
constexpr int kTuneNamed = (int)(sizeof(kTuneNames) / sizeof(kTuneNames[0]));
static_assert(kTuneNamed <= FNSSL_TUNE_COUNT, "more knob names than slots");
static_assert(kTuneNamed == FNSSL_TUNE_NO_STATIC4 + 1, "knob names out of step with include/fnssl.h");
}  // namespace

const fnssl_tuning& tuning() {
  if (g_call_tuning) return *g_call_tuning;
  const unsigned v = g_tuning_version.load(std::memory_order_acquire);
  if (v != g_thread_version) {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    g_thread_tuning = g_default_tuning;
    g_thread_version = g_tuning_version.load(std::memory_order_relaxed);
  }
  return g_thread_tuning;
}

TuningScope::TuningScope(const fnssl_tuning* t) : prev_(g_call_tuning), active_(false) {
  if (t && t->struct_bytes == sizeof(fnssl_tuning)) {
    g_call_tuning = t;
    active_ = true;
  }
}
TuningScope::~TuningScope() {
  if (active_) g_call_tuning = prev_;
}

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
  if (!g_timing || !name) return;
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

extern "C" int fnssl_tuning_set(const fnssl_tuning* t) {
  FNSSL_REQUIRE(!t || t->struct_bytes == sizeof(fnssl_tuning), "tuning_set: struct_bytes %u, this library's fnssl_tuning has %zu",
                t ? t->struct_bytes : 0u, sizeof(fnssl_tuning));
  std::lock_guard<std::mutex> lk(fnssl::g_tuning_mu);
  fnssl::g_default_tuning = t ? *t : fnssl_tuning{sizeof(fnssl_tuning), {0}};
  fnssl::g_tuning_version.fetch_add(1, std::memory_order_release);
  return FNSSL_OK;
}

extern "C" int fnssl_tuning_get(fnssl_tuning* t) {
  FNSSL_REQUIRE(t != nullptr, "tuning_get: NULL");
  std::lock_guard<std::mutex> lk(fnssl::g_tuning_mu);
  *t = fnssl::g_default_tuning;
  return FNSSL_OK;
}

namespace {
// What another tenant's persistent kernels do to the CUs (RCCL's all-reduce kernels under an overlapped backward): each
// workgroup claims `lds` bytes of LDS (160 KiB = the whole CU) and idles until *stop != 0 or the time limit.
__global__ void __launch_bounds__(64) occupy_kernel(unsigned* stop, long long max_ticks) {
  extern __shared__ char occ_smem[];
  if (threadIdx.x == 0) {
    occ_smem[0] = 1;   // the allocation is real
    if (stop) __hip_atomic_store(stop + 1 + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // "I am resident"
  }
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < max_ticks) {
    if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
    __builtin_amdgcn_s_sleep(64);
  }
}
}  // namespace

extern "C" int fnssl_occupy_cus(int nblocks, int lds_bytes, unsigned* stop, int max_ms, void* stream) {
  FNSSL_REQUIRE(nblocks > 0 && nblocks <= 4096 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && max_ms > 0 && max_ms <= 60000,
                "occupy_cus: nblocks %d, lds %d B, limit %d ms", nblocks, lds_bytes, max_ms);
  if (lds_bytes > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  int rate_khz = 100000;   // wall_clock64 ticks: the constant "wall clock" counter, 100 MHz on gfx9
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);
  if (rate_khz <= 0) rate_khz = 100000;
  hipLaunchKernelGGL(occupy_kernel, dim3(nblocks), dim3(64), (size_t)lds_bytes, fnssl::as_stream(stream), stop,
                     (long long)max_ms * rate_khz);
  FNSSL_CHECK_LAUNCH("occupy_kernel");
  return FNSSL_OK;
}

extern "C" const char* fnssl_tuning_name(int index) {
  return index >= 0 && index < fnssl::kTuneNamed ? fnssl::kTuneNames[index] : nullptr;
}

namespace {
typedef float peak_v4f __attribute__((ext_vector_type(4)));
// The box's own fp32-MFMA ceiling: nothing but v_mfma_f32_16x16x4_f32 on four independent accumulators per wave
// (dependent latency 40 cycles < 4 x 32 cycles of issue), `wps` waves per SIMD on every CU.  bench.py times it with
// HIP events and reports it as roofline.peak_measured, so that a 2 % difference between two boxes (clock, power
// budget) cannot hide or fake a 2 % kernel gain.
// `clk` (may be null): per workgroup {shader cycles, 100 MHz wall ticks, XCC id} of wave 0 over the whole loop — the clock the
// XCD actually ran at while the matrix pipe was saturated (the sustained mode of bench.py reads it after >= 2 s of launches).
__global__ void __launch_bounds__(256) mfma_f32_peak_kernel(float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  unsigned long long c0 = 0, r0 = 0;
  if (clk) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  peak_v4f acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = peak_v4f{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + 0.001f * lane, b = 0.5f - 0.002f * lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[3], 0, 0, 0);
    }
    asm volatile("" : "+v"(a), "+v"(b));   // keep the loop a loop
  }
  const peak_v4f r = acc[0] + acc[1] + acc[2] + acc[3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r.x + r.y + r.z + r.w;
  if (clk && threadIdx.x == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    clk[3 * (size_t)blockIdx.x + 0] = c1 - c0;
    clk[3 * (size_t)blockIdx.x + 1] = r1 - r0;
    clk[3 * (size_t)blockIdx.x + 2] = xcc & 15u;
  }
}
}  // namespace

extern "C" {

int fnssl_mfma_f32_peak_clocks(float* out, size_t out_floats, int iters, int waves_per_simd, double* flop,
                               unsigned long long* clocks, size_t clocks_len, void* stream) {
  FNSSL_REQUIRE(out && flop && iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 8, "mfma_f32_peak: bad arguments");
  const int nblk = fnssl::device_cus() * waves_per_simd;
  FNSSL_REQUIRE(out_floats >= (size_t)nblk * 256, "mfma_f32_peak: out needs %zu floats", (size_t)nblk * 256);
  FNSSL_REQUIRE(!clocks || clocks_len >= (size_t)nblk * 3, "mfma_f32_peak: clocks needs %zu entries", (size_t)nblk * 3);
  hipLaunchKernelGGL(mfma_f32_peak_kernel, dim3(nblk), dim3(256), 0, fnssl::as_stream(stream), out, iters, clocks);
  FNSSL_CHECK_LAUNCH("mfma_f32_peak_kernel");
  *flop = (double)nblk * 4 * (double)iters * 64 * (2.0 * 16 * 16 * 4);
  return FNSSL_OK;
}

int fnssl_mfma_f32_peak(float* out, size_t out_floats, int iters, int waves_per_simd, double* flop, void* stream) {
  return fnssl_mfma_f32_peak_clocks(out, out_floats, iters, waves_per_simd, flop, nullptr, 0, stream);
}

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
