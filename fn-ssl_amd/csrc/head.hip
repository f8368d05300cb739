// DP-IPD head: 12-frame mean -> Linear(256, 2) -> tanh -> [cos | sin] packing
// (reference FN-SSL/Model.py:67-69,79-87) and the optional ipd2doa Linear
// (Model.py:70-71,88-89).  HBM-bound: the head reads the last narrow-band
// output once (1 KiB rows, one wave per (pair, bin, segment), float4 per lane)
// and reduces 256 channels with wavefront shuffles.
#include "common.h"

namespace {

constexpr int kSeg = FNSSL_SEG_FRAMES;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// x [nb, nf, nt, 256] -> out [nb, nt2, 2*nf]
__global__ void __launch_bounds__(256)
head_kernel(const float* __restrict__ x, int nb, int nf, int nt, int nt2, const float* __restrict__ w,
            const float* __restrict__ bias, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, f, seg)
  const long long nitems = (long long)nb * nf * nt2;
  if (item >= nitems) return;
  const int seg = (int)(item % nt2);
  const int f = (int)((item / nt2) % nf);
  const int b = (int)(item / ((long long)nt2 * nf));
  const float4* row = reinterpret_cast<const float4*>(x + (((long long)b * nf + f) * nt + (long long)seg * kSeg) * 256) + lane;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < kSeg; ++r) {
    const float4 v = row[r * 64];
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  const float inv = (float)kSeg;
  s.x = __fdiv_rn(s.x, inv);   // AvgPool2d divides the window sum
  s.y = __fdiv_rn(s.y, inv);
  s.z = __fdiv_rn(s.z, inv);
  s.w = __fdiv_rn(s.w, inv);
  const float4 w0 = reinterpret_cast<const float4*>(w)[lane];
  const float4 w1 = reinterpret_cast<const float4*>(w + 256)[lane];
  float d0 = s.x * w0.x + s.y * w0.y + s.z * w0.z + s.w * w0.w;
  float d1 = s.x * w1.x + s.y * w1.y + s.z * w1.z + s.w * w1.w;
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  if (lane == 0) {
    float* o = out + ((long long)b * nt2 + seg) * (2 * nf);
    o[f] = tanhf(d0 + bias[0]);
    o[nf + f] = tanhf(d1 + bias[1]);
  }
}

// y[m, n] = sum_k x[m, k] wt[k, n] + b[n]; one block per row, one thread per output
__global__ void linear_kernel(const float* __restrict__ x, int k, const float* __restrict__ wt,
                              const float* __restrict__ b, int n_out, float* __restrict__ y) {
  extern __shared__ float xs[];
  const long long row = blockIdx.x;
  for (int i = threadIdx.x; i < k; i += blockDim.x) xs[i] = x[row * k + i];
  __syncthreads();
  for (int n = threadIdx.x; n < n_out; n += blockDim.x) {
    float acc = b ? b[n] : 0.f;
    for (int i = 0; i < k; ++i) acc = fmaf(xs[i], wt[(long long)i * n_out + n], acc);
    y[row * n_out + n] = acc;
  }
}

}  // namespace

extern "C" {

int fnssl_head(const float* x, int nb, int nf, int nt, const float* w, const float* b, float* out,
               void* stream) {
  FNSSL_REQUIRE(x && w && b && out, "head: null pointer");
  FNSSL_REQUIRE(nb > 0 && nf > 0 && nt > 0, "head: empty problem");
  const int nt2 = nt / kSeg;
  if (nt2 == 0) return FNSSL_OK;   // fewer than 12 frames: empty output, like AvgPool2d's floor
  const long long nitems = (long long)nb * nf * nt2;
  const long long nblk = (nitems + 3) / 4;
  FNSSL_REQUIRE(nblk < (1ll << 31), "head: too many items");
  fnssl::TimedLaunch tl("head", fnssl::as_stream(stream));
  hipLaunchKernelGGL(head_kernel, dim3((unsigned)nblk), dim3(256), 0, fnssl::as_stream(stream), x, nb, nf, nt,
                     nt2, w, b, out);
  FNSSL_CHECK_LAUNCH("head_kernel");
  return FNSSL_OK;
}

int fnssl_linear(const float* x, int m, int k, const float* wt, const float* b, int n_out, float* y,
                 void* stream) {
  FNSSL_REQUIRE(x && wt && y, "linear: null pointer");
  FNSSL_REQUIRE(m >= 0 && k > 0 && n_out > 0 && k <= 8192, "linear: bad shape (m %d, k %d, n %d)", m, k, n_out);
  if (m == 0) return FNSSL_OK;
  fnssl::TimedLaunch tl("linear", fnssl::as_stream(stream));
  hipLaunchKernelGGL(linear_kernel, dim3(m), dim3(192), k * sizeof(float), fnssl::as_stream(stream), x, k, wt, b,
                     n_out, y);
  FNSSL_CHECK_LAUNCH("linear_kernel");
  return FNSSL_OK;
}

}  // extern "C"
