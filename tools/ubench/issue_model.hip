// Micro-benchmark (round 4): what does an instruction cost BESIDE v_mfma_f32_16x16x4_f32 on gfx950?
//
// The fp32 LSTM kernels run at 0.83-0.87 of the fp32 MFMA roof although every ablation (profiles/r04) shows that no
// single memory / LDS / gate component bounds them: removing ALL of them still leaves 0.91, and the gate math costs as
// much as if it were serialized with the matrix instructions.  This program measures, for W waves per SIMD:
//   mode A: after every group of 4 MFMAs (4 independent accumulators), F filler instructions of one kind;
//   mode B: 96 groups of 4 MFMAs, then the same 96 F fillers as one block (the "gate phase" of the kernels).
// and prints the time per MFMA in cycles of the box's own calibrated MFMA clock (= the 0-filler run at 32 cycles per
// MFMA) and the cost per filler instruction.
// Build: hipcc --offload-arch=gfx950 -O3 issue_model.hip -o issue_model ; run: ./issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                                \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

template <int I, int N, class Fn>
__device__ __forceinline__ void rep(Fn&& f) {
  if constexpr (I < N) {
    f();
    rep<I + 1, N>(f);
  }
}

// KIND: 0 none, 1 v_fma_f32, 2 v_exp_f32, 3 v_add_u32, 4 s_add_u32, 5 ds_read_b128, 6 v_pk_fma_f32, 7 s_nop 0,
//       8 v_mul_f32 + v_exp_f32 + v_add_f32 + v_rcp_f32 (a sigmoid: F counts sigmoids), 9 buffer/global load dwordx4
template <int KIND, int F, int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  v4f acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + 0.001f * lane, b = 0.5f - 0.002f * lane;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.25f + 0.01f * i + 0.001f * lane;
  v2f px[4];
  for (int i = 0; i < 4; ++i) px[i] = v2f{0.5f + i, 0.25f + lane};
  unsigned u[4] = {1u + lane, 2u, 3u, 4u};
  unsigned sc = 0;
  v4f ld[4];
  for (int i = 0; i < 4; ++i) ld[i] = v4f{0.f, 0.f, 0.f, 0.f};
  const char* lds = smem + lane * 16;
  const float* gp = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
  int slot = 0;
  auto filler = [&]() {
    const int s = slot & 3;
    if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[slot & 7]) : "v"(a), "v"(b));
    if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[slot & 7]));
    if constexpr (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[s]) : "v"(u[(s + 1) & 3]));
    if constexpr (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
    if constexpr (KIND == 5) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(ld[s]) : "v"((unsigned)(size_t)lds + 1024u * s));
    if constexpr (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(px[s]) : "v"(px[(s + 1) & 3]));
    if constexpr (KIND == 7) asm volatile("s_nop 0");
    if constexpr (KIND == 8)
      asm volatile("v_mul_f32 %0, %0, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0" : "+v"(x[slot & 7]) : "v"(b));
    if constexpr (KIND == 9) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[s]) : "v"(gp));
    ++slot;
  };
  auto mfma4 = [&]() {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[3], 0, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
      // 8 groups per loop trip, fillers right behind each group
      rep<0, 8>([&]() {
        mfma4();
        __builtin_amdgcn_sched_barrier(0);
        rep<0, F>(filler);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (KIND == 5 || KIND == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      // phase structure: 96 groups, then the block of 96 F fillers (in 12 sub-blocks so that code size stays sane)
      for (int g = 0; g < 12; ++g) {
        rep<0, 8>([&]() { mfma4(); });
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int g = 0; g < 12; ++g) {
        rep<0, 8 * F>(filler);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND == 5 || KIND == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
    asm volatile("" : "+v"(a), "+v"(b));
  }
  v4f r = acc[0] + acc[1] + acc[2] + acc[3] + ld[0] + ld[1] + ld[2] + ld[3];
  float t = r.x + r.y + r.z + r.w;
  for (int i = 0; i < 8; ++i) t += x[i];
  for (int i = 0; i < 4; ++i) t += px[i].x + px[i].y + (float)u[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = t + (float)sc;
}

static float* g_out;
static float* g_src;
static int g_ncu;

template <int KIND, int F, int MODE>
double run(int wps, int iters) {
  const int nblk = g_ncu * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<KIND, F, MODE>), dim3(nblk), dim3(256), 16384, 0, g_out, g_src, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  CK(hipGetLastError());
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  // ns per MFMA per SIMD: every SIMD runs wps waves x iters x (MODE 0: 8 groups; MODE 1: 96 groups) x 4 MFMAs
  const double mf = (double)wps * iters * (MODE == 0 ? 8 : 96) * 4;
  return best * 1e6 / mf;
}

static double g_ns32 = 0;   // ns of one 32-cycle MFMA slot on this box (calibrated by the filler-free run)

template <int KIND, int F>
void row(const char* name) {
  for (int mode = 0; mode < 2; ++mode) {
    printf("%-14s F=%d mode %c:", name, F, mode ? 'B' : 'A');
    for (int wps : {1, 2, 4}) {
      const int iters = mode ? 40 : 480;
      const double ns = mode ? run<KIND, F, 1>(wps, iters) : run<KIND, F, 0>(wps, iters);
      const double cyc = ns / g_ns32 * 32.0;                 // cycles per MFMA slot
      const double per_filler = (cyc - 32.0) * 4.0 / F;      // F fillers per 4 MFMAs
      printf("   W=%d %6.2f cyc/MFMA (%5.2f per filler)", wps, cyc, per_filler);
    }
    printf("\n");
    fflush(stdout);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_ncu = prop.multiProcessorCount;
  CK(hipMalloc(&g_out, (size_t)g_ncu * 8 * 256 * sizeof(float)));
  CK(hipMalloc(&g_src, (size_t)g_ncu * 8 * 256 * 4 * sizeof(float)));
  CK(hipMemset(g_src, 0, (size_t)g_ncu * 8 * 256 * 4 * sizeof(float)));
  printf("device %s, %d CUs\n", prop.name, g_ncu);
  for (int wps : {1, 2, 4}) {
    const double a = run<0, 0, 0>(wps, 480), b = run<0, 0, 1>(wps, 40);
    printf("no fillers     W=%d: %.3f / %.3f ns per MFMA (mode A / B)  -> %.1f TFLOP/s\n", wps, a, b,
           2.0 * 16 * 16 * 4 / a * 1e-3 * g_ncu * 4);
    if (wps == 1) g_ns32 = a;
  }
  row<1, 2>("v_fma_f32");
  row<1, 4>("v_fma_f32");
  row<1, 8>("v_fma_f32");
  row<2, 2>("v_exp_f32");
  row<2, 4>("v_exp_f32");
  row<3, 4>("v_add_u32");
  row<3, 8>("v_add_u32");
  row<4, 4>("s_add_u32");
  row<4, 8>("s_add_u32");
  row<5, 1>("ds_read_b128");
  row<5, 2>("ds_read_b128");
  row<6, 2>("v_pk_fma_f32");
  row<6, 4>("v_pk_fma_f32");
  row<7, 4>("s_nop");
  row<7, 8>("s_nop");
  row<8, 1>("sigmoid(4)");
  row<8, 2>("sigmoid(4)");
  row<9, 1>("global_load4");
  return 0;
}
