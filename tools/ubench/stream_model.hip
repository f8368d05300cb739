// Micro-benchmark (round 4): issue efficiency of the fp32 LSTM kernels' MATRIX PHASE as written in HIP.
//
// Finding it follows up (profiles/r04/f32c_timeline.txt): with every load, store, tag and gate instruction ablated,
// lstm_f32c_kernel's waves still need 187.1 M shader cycles for 176.2 M cycles of v_mfma_f32_16x16x4_f32 (94.2 %), while a
// bare MFMA loop reaches 98.6 %.  The matrix phase is: per "quad" 16 MFMAs (4 gates x 4 k-steps) whose A operands are four
// 1-KiB LDS records (ds_read_b128 per lane) and whose B operands are 4 registers.  This program runs that stream — and
// nothing else — in several schedules and reports shader cycles per MFMA of the slowest wave (s_memtime inside the kernel:
// independent of clock, launch overhead and XCD), for 3 and 4 waves per SIMD:
//   V0  lstm_f32c.h's quad: record j of quad Q + 1 is read right behind the MFMAs that used record j of quad Q (fenced)
//   V1  lstm_static.h's SQUAD: records 2, 3 at the quad's start, 0, 1 of the next quad in its middle (fenced)
//   V2  V0 without scheduling fences (the compiler places the reads)
//   V3  whole next quad read at the quad's start into a second register set (4 reads back to back, one wait per quad)
//   V4  two quads per round: 8 reads back to back for the round after next, 32 MFMAs
//   V5  V0 with the accumulators re-initialised from registers instead of LDS records (no reads at the group-step boundary)
// Build: hipcc --offload-arch=gfx950 -O3 stream_model.hip -o stream_model ; run: ./stream_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                       \
    }                                                                \
  } while (0)

template <int I>
using ic = std::integral_constant<int, I>;
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<N, I + 1>(f);
  }
}

#define MFMA4(ACC, AV, BV)                                                          \
  do {                                                                              \
    ACC[0] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).x, (BV), ACC[0], 0, 0, 0);   \
    ACC[1] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).y, (BV), ACC[1], 0, 0, 0);   \
    ACC[2] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).z, (BV), ACC[2], 0, 0, 0);   \
    ACC[3] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).w, (BV), ACC[3], 0, 0, 0);   \
  } while (0)

constexpr int QPS = 25;   // quads per group-step (bias quad + 24), as the 256-channel full-band layer

template <int V, int NW>
__global__ void __launch_bounds__(NW * 64) k(float* out, unsigned long long* cyc, int gsteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < QPS * 4 * 64; i += NW * 64) reinterpret_cast<v4f*>(smem)[i] = v4f{1e-3f * (i & 7), 2e-3f, -1e-3f, 5e-4f};
  __syncthreads();
  const char* const lds_rd = smem + lane * 16;
  auto rec = [&](int q, int j) { return *reinterpret_cast<const v4f*>(lds_rd + (q * 4 + j) * 1024); };
  v4f bop[8];
  for (int i = 0; i < 8; ++i) bop[i] = v4f{0.01f * i, 0.02f + lane * 1e-3f, -0.01f, 0.03f};
  v4f sum = v4f{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int gs = 0; gs < gsteps; ++gs) {
    v4f acc[4];
    if constexpr (V == 5) {
      static_for<4>([&](auto j) { acc[j.value] = bop[j.value]; });
    } else {
      static_for<4>([&](auto j) { acc[j.value] = rec(0, j.value); });
    }
    if constexpr (V == 0 || V == 2 || V == 5) {
      v4f ra[4];
      static_for<4>([&](auto j) { ra[j.value] = rec(1, j.value); });
      static_for<QPS - 1>([&](auto qc) {
        constexpr int Q = 1 + decltype(qc)::value;
        const v4f b = bop[Q & 7];
        static_for<4>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          MFMA4(acc, ra[J], b[J]);
          if constexpr (V != 2) __builtin_amdgcn_sched_barrier(0);
          if constexpr (Q + 1 < QPS) ra[J] = rec(Q + 1, J);
          if constexpr (V != 2) __builtin_amdgcn_sched_barrier(0);
        });
      });
    } else if constexpr (V == 1) {
      v4f a0 = rec(1, 0), a1 = rec(1, 1);
      static_for<QPS - 1>([&](auto qc) {
        constexpr int Q = 1 + decltype(qc)::value;
        const v4f b = bop[Q & 7];
        const v4f a2 = rec(Q, 2), a3 = rec(Q, 3);
        __builtin_amdgcn_sched_barrier(0);
        MFMA4(acc, a0, b.x);
        MFMA4(acc, a1, b.y);
        if constexpr (Q + 1 < QPS) {
          a0 = rec(Q + 1, 0);
          a1 = rec(Q + 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        MFMA4(acc, a2, b.z);
        MFMA4(acc, a3, b.w);
      });
    } else if constexpr (V == 3) {
      v4f ra[2][4];
      static_for<4>([&](auto j) { ra[1][j.value] = rec(1, j.value); });
      static_for<QPS - 1>([&](auto qc) {
        constexpr int Q = 1 + decltype(qc)::value;
        const v4f b = bop[Q & 7];
        if constexpr (Q + 1 < QPS) static_for<4>([&](auto j) { ra[(Q + 1) & 1][j.value] = rec(Q + 1, j.value); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto jc) { MFMA4(acc, ra[Q & 1][decltype(jc)::value], b[decltype(jc)::value]); });
        __builtin_amdgcn_sched_barrier(0);
      });
    } else if constexpr (V == 4) {
      v4f ra[2][8];
      static_for<8>([&](auto j) { ra[0][j.value] = rec(1 + j.value / 4, j.value & 3); });
      static_for<(QPS - 1) / 2>([&](auto rc) {
        constexpr int R = decltype(rc)::value, Q = 1 + 2 * R;
        if constexpr (Q + 2 < QPS) static_for<8>([&](auto j) { ra[(R + 1) & 1][j.value] = rec(Q + 2 + j.value / 4, j.value & 3); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          MFMA4(acc, ra[R & 1][J], bop[(Q + J / 4) & 7][J & 3]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    sum += acc[0] + acc[1] + acc[2] + acc[3];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * NW + w] = c1 - c0;
  out[(size_t)blockIdx.x * NW * 64 + threadIdx.x] = sum.x + sum.y + sum.z + sum.w;
}

template <int V, int NW>
void run(float* out, unsigned long long* cyc, int ncu, const char* name) {
  const int gsteps = 400;
  const size_t lds = (size_t)QPS * 4096;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<V, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<unsigned long long> h((size_t)ncu * NW);
  double best = 1e30, avg = 0;
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL((k<V, NW>), dim3(ncu), dim3(NW * 64), lds, 0, out, cyc, gsteps);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
    // a SIMD hosts waves w, w + 4, ..: it is done when its last wave is; per SIMD MFMAs = (NW / 4) waves x gsteps x 24 x 16
    double worst = 0, mean = 0;
    for (int b = 0; b < ncu; ++b)
      for (int s = 0; s < 4; ++s) {
        unsigned long long m = 0;
        for (int w = s; w < NW; w += 4) m = std::max(m, h[(size_t)b * NW + w]);
        worst = std::max(worst, (double)m);
        mean += (double)m;
      }
    mean /= ncu * 4;
    const double mf = (double)(NW / 4) * gsteps * (QPS - 1) * 16;
    if (r > 0 && mean / mf < best) {
      best = mean / mf;
      avg = worst / mf;
    }
  }
  printf("%-46s %2d waves: %6.2f cycles per MFMA (mean over SIMDs; worst SIMD %6.2f) -> %.1f %% of the 32-cycle pipe\n", name, NW,
         best, avg, 3200.0 / best);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)ncu * 1024 * sizeof(float)));
  CK(hipMalloc(&cyc, (size_t)ncu * 16 * sizeof(unsigned long long)));
#define ROW(V, NAME) \
  run<V, 16>(out, cyc, ncu, NAME); \
  run<V, 12>(out, cyc, ncu, NAME); \
  run<V, 8>(out, cyc, ncu, NAME);  \
  run<V, 4>(out, cyc, ncu, NAME);
  ROW(0, "V0 f32c quad (read behind use, fenced)")
  ROW(1, "V1 static SQUAD (half-quad peeks, fenced)")
  ROW(2, "V2 as V0, no fences")
  ROW(3, "V3 next quad's 4 reads at the quad start")
  ROW(4, "V4 two quads per round, 8 reads at once")
  ROW(5, "V5 as V0, accumulators from registers")
  return 0;
}
