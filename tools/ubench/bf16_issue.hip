// Micro-benchmark (round 6): what bounds the bf16 LSTM kernels of config 3 (lstm_bf16c.h) on gfx950?
//
// Their counters (profiles/r06/a_pmc_c3_*.json): MFMA pipe busy 0.54 (narrow-band) / 0.33 / 0.24 (full-band), VALU issue
// 0.17 of the wave cycles per wave at two waves per SIMD, no LDS bank conflicts, and a shader clock of ~1.75 GHz under
// that load.  Questions this program answers with nothing but v_mfma_f32_32x32x16_bf16 and the gate arithmetic's
// instruction mix (per 32 x 32 tile of gate rows: 136 MFMAs of a part against ~72 VALU instructions per tile x 4 tiles, 40 of
// every 72 transcendental):
//   1. the sustained bf16-MFMA rate of THIS box and the clock it holds (1 and 2 waves per SIMD, operands in registers;
//      with one ds_read_b128 per MFMA like the kernel's A operands);
//   2. gate arithmetic interleaved into the MFMA stream of the SAME wave (F sigmoids = mul, exp, add, rcp after every MFMA);
//   3. the kernel's structure: two waves per SIMD, each alternating a matrix phase (136 MFMAs) and a gate phase
//      (4 x 40 transcendentals + 4 x 32 others), free-running — how much of the gate phase hides under the partner's MFMAs;
//   4. the same work with wave 0 of a SIMD doing ONLY MFMAs and wave 1 ONLY gate arithmetic (perfect overlap if the two
//      pipes were independent).
// Prints ms, TFLOP/s, shader cycles per MFMA (s_memtime) and the clock (s_memtime / s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 bf16_issue.hip -o bf16_issue ; run: ./bf16_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

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
    f(I);
    rep<I + 1, N>(f);
  }
}

__device__ __forceinline__ void sigmoid_asm(float& x) {   // the gate arithmetic's unit: v_mul, v_exp, v_add, v_rcp
  asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %0\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0" : "+v"(x));
}

// MODE 0: MFMAs only.  1: + one ds_read_b128 per MFMA (A operand).  2: + F sigmoids after every MFMA (same wave).
// 3: phases — PM MFMAs, then PG sigmoids (all waves, free-running).  4: even waves only MFMAs, odd waves only sigmoids
// (PM MFMAs / PG sigmoids per round each).
template <int MODE, int F>
__global__ void __launch_bounds__(512) k(float* out, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v16f acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  v8bf a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)(0.001f * (lane + j));
    b[j] = (__bf16)(0.5f - 0.002f * (lane - j));
  }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.25f + 0.01f * i + 0.001f * lane;
  const char* lds = smem + lane * 16;
  constexpr int PM = 136, PG = 160;   // a part: 34 K-steps x 4 tiles; its gate phase: 4 tiles x 40 transcendentals = 160 sigmoid units of 4 instructions (640 VALU, the kernel has ~290 + waits)
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE <= 2) {
      rep<0, 16>([&](int u) {
        v8bf av = a;
        if constexpr (MODE == 1) av = __builtin_bit_cast(v8bf, *reinterpret_cast<const v4f*>(lds + (u & 7) * 1024));
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[u & 3], 0, 0, 0);
        if constexpr (MODE == 2) rep<0, F>([&](int f) { sigmoid_asm(x[(u + f) & 7]); });
      });
    } else if constexpr (MODE == 3) {
      rep<0, PM>([&](int u) { acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0); });
      rep<0, F>([&](int u) { sigmoid_asm(x[u & 7]); });
    } else {
      if ((wave & 4) == 0) {   // waves 0-3: one per SIMD, matrix only
        rep<0, PM>([&](int u) { acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0); });
      } else {                 // waves 4-7: their SIMD partners, gate arithmetic only
        rep<0, F>([&](int u) { sigmoid_asm(x[u & 7]); });
      }
    }
    asm volatile("" : "+v"(a), "+v"(b));
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  v16f r = acc[0] + acc[1] + acc[2] + acc[3];
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += r[j];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (clk && lane == 0) {
    clk[2 * ((size_t)blockIdx.x * (blockDim.x / 64) + wave) + 0] = c1 - c0;
    clk[2 * ((size_t)blockIdx.x * (blockDim.x / 64) + wave) + 1] = r1 - r0;
  }
  (void)PG;
}

template <int MODE, int F>
static void run(const char* what, int waves_per_simd, double mfma_per_iter_per_wave, double seconds = 0.4) {
  int dev = 0, ncu = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const int threads = 256 * waves_per_simd, nw = ncu * 4 * waves_per_simd;
  float* out;
  unsigned long long* clk;
  CK(hipMalloc(&out, (size_t)ncu * threads * sizeof(float)));
  CK(hipMalloc(&clk, (size_t)nw * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto kern = k<MODE, F>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  int iters = 200;
  float ms = 0.f;
  for (int pass = 0; pass < 2; ++pass) {   // pass 0 sizes the launch for ~`seconds`
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(ncu), dim3(threads), 64 * 1024, 0, out, iters, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) iters = (int)(iters * seconds * 1e3 / (ms > 1e-3f ? ms : 1e-3f)) + 1;
  }
  std::vector<unsigned long long> h((size_t)nw * 2);
  CK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0, mhz_lo = 1e9, mhz_hi = 0;
  for (int i = 0; i < nw; ++i) {
    cyc += (double)h[2 * i];
    const double mhz = (double)h[2 * i] / (double)(h[2 * i + 1] ? h[2 * i + 1] : 1) * 100.0;
    mhz_lo = mhz < mhz_lo ? mhz : mhz_lo;
    mhz_hi = mhz > mhz_hi ? mhz : mhz_hi;
  }
  cyc /= nw;
  // MFMAs per SIMD: in mode 4 only half the waves multiply
  const double mf_waves = MODE == 4 ? 0.5 * waves_per_simd : waves_per_simd;
  const double mfma_per_simd = mfma_per_iter_per_wave * iters * mf_waves;
  const double flop = mfma_per_simd * ncu * 4 * 32768.0;
  printf("%-86s %7.2f ms %8.1f TFLOP/s  %6.1f cyc/MFMA/SIMD  clock %4.0f-%4.0f MHz\n", what, ms, flop / (ms * 1e-3) / 1e12,
         cyc / mfma_per_simd, mhz_lo, mhz_hi);
  CK(hipFree(out));
  CK(hipFree(clk));
}

int main() {
  printf("v_mfma_f32_32x32x16_bf16, one workgroup per CU; 'cyc/MFMA/SIMD' = wave cycles / MFMAs issued on that SIMD (32 = pipe full)\n");
  run<0, 0>("1 wave/SIMD, MFMA only", 1, 16);
  run<0, 0>("2 waves/SIMD, MFMA only", 2, 16);
  run<0, 0>("2 waves/SIMD, MFMA only, 2 s sustained", 2, 16, 2.0);
  run<1, 0>("2 waves/SIMD, + ds_read_b128 per MFMA", 2, 16);
  run<2, 1>("1 wave/SIMD, + 1 sigmoid (4 VALU, 2 transcendental) after every MFMA, same wave", 1, 16);
  run<2, 2>("1 wave/SIMD, + 2 sigmoids after every MFMA, same wave", 1, 16);
  run<2, 1>("2 waves/SIMD, + 1 sigmoid after every MFMA", 2, 16);
  run<2, 2>("2 waves/SIMD, + 2 sigmoids after every MFMA", 2, 16);
  run<3, 0>("2 waves/SIMD, phases: 136 MFMAs, no gate phase", 2, 136);
  run<3, 80>("2 waves/SIMD, phases: 136 MFMAs then 80 sigmoids (320 VALU), free-running", 2, 136);
  run<3, 160>("2 waves/SIMD, phases: 136 MFMAs then 160 sigmoids (640 VALU), free-running", 2, 136);
  run<3, 80>("1 wave/SIMD, phases: 136 MFMAs then 80 sigmoids", 1, 136);
  run<4, 80>("2 waves/SIMD, wave A only MFMAs (136), wave B only sigmoids (80) per round", 2, 136);
  run<4, 160>("2 waves/SIMD, wave A only MFMAs (136), wave B only sigmoids (160) per round", 2, 136);
  run<4, 320>("2 waves/SIMD, wave A only MFMAs (136), wave B only sigmoids (320) per round", 2, 136);
  return 0;
}
