// Micro-benchmark: achievable v_mfma_f32_16x16x4_f32 rate on gfx950 as a function of waves per
// SIMD and of what else shares the instruction stream (LDS A-operand reads, scalar bookkeeping).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, A/B in registers; 1: + ds_read_b128 of A per 4 MFMAs; 2: + scalar counter/branch per 16
__global__ void __launch_bounds__(1024) k(float* out, int iters, int nquads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v4f a = {1.f + lane, 2.f, 3.f, 4.f};
  float b = 0.5f + lane;
  const char* rd = smem + lane * 16;
  int left = 11, rq = 0;
  for (int it = 0; it < iters; ++it) {
    for (int q = 0; q < nquads; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v4f av = a;
        if (MODE >= 1) av = *reinterpret_cast<const v4f*>(rd + rq * 4096 + j * 1024);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b, acc[3], 0, 0, 0);
      }
      if (MODE >= 2) {
        rq = (rq + 1 == 22) ? 0 : rq + 1;
        if (--left == 0) {
          left = 11;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
  }
  v4f r = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y + r.z + r.w;
}

// mode 3/4/5: the kernel's software-pipelined quad (A operands peeked half a quad ahead).
//   3: ring bookkeeping, no barrier   4: + s_barrier at chunk ends   5: as 3 without sched_barrier pins
template <int MODE>
__global__ void __launch_bounds__(1024) kq(float* out, int iters, int nquads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v4f hold[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) hold[i] = v4f{0.5f + lane + i, 1.f, 2.f + i, 3.f};
  const char* rd = smem + lane * 16;
  int left = 11, rq = 0;
  auto rec = [&](int q, int j) { return *reinterpret_cast<const v4f*>(rd + q * 4096 + j * 1024); };
  v4f a0 = rec(0, 0), a1 = rec(0, 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int sp = 0; sp < 16; ++sp) {
      const v4f a2 = rec(rq, 2), a3 = rec(rq, 3);
      if (MODE != 5) __builtin_amdgcn_sched_barrier(0);
#define M4(A, B)                                                              \
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B), acc[0], 0, 0, 0); \
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B), acc[1], 0, 0, 0); \
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B), acc[2], 0, 0, 0); \
  acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B), acc[3], 0, 0, 0);
      M4(a0, hold[sp].x) M4(a1, hold[sp].y)
      const int nq = (rq + 1 == 22) ? 0 : rq + 1;
      a0 = rec(nq, 0);
      a1 = rec(nq, 1);
      if (MODE != 5) __builtin_amdgcn_sched_barrier(0);
      M4(a2, hold[sp].z) M4(a3, hold[sp].w)
      rq = nq;
      if (--left == 0) {
        left = 11;
        if (MODE == 4)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a0 = rec(rq, 0);
        a1 = rec(rq, 1);
      }
    }
  }
  v4f r = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y + r.z + r.w;
}

template <int MODE>
void runq(int waves_per_simd, float* d) {
  const int threads = waves_per_simd * 4 * 64;
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kq<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(kq<MODE>, dim3(256), dim3(threads), 96 * 1024, 0, d, 10, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kq<MODE>, dim3(256), dim3(threads), 96 * 1024, 0, d, iters, 16);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * waves_per_simd * 4 * (double)iters * 16 * 16 * 2048.0;
  printf("mode %d  waves/SIMD %d : %7.2f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)\n", MODE, waves_per_simd, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

template <int MODE>
void run(int waves_per_simd, float* d) {
  const int threads = waves_per_simd * 4 * 64;   // one workgroup per CU
  const int iters = 2000, nquads = 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 96 * 1024, 0, d, 10, nquads);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 96 * 1024, 0, d, iters, nquads);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * waves_per_simd * 4 * (double)iters * nquads * 16 * 2048.0;
  printf("mode %d  waves/SIMD %d : %7.2f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)\n", MODE, waves_per_simd, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * sizeof(float));
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int w = 1; w <= 4; ++w) run<0>(w, d);
  for (int w = 1; w <= 4; ++w) run<1>(w, d);
  for (int w = 1; w <= 4; ++w) run<2>(w, d);
  for (int w = 1; w <= 4; ++w) runq<3>(w, d);
  for (int w = 1; w <= 4; ++w) runq<4>(w, d);
  for (int w = 1; w <= 4; ++w) runq<5>(w, d);
  hipFree(d);
  return 0;
}
