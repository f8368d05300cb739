// IPD -> DOA back end on device: iterative source detection and localisation
// (reference SourceDetectLocalize.forward, meth_mode 'IDL',
// FN-SSL/Lightning/Module.py:525-577, as driven by PredDOA.predgt2DOA :690-727).
//
// For every (utterance, segment) the reference does a bmm against the template bank,
// an argmax, then a Python double loop that gathers the winning template, projects
// and subtracts it — bouncing to numpy per frame.  Here one workgroup owns one
// (utterance, segment): the 2nf*np-vector lives in LDS, each wave scores a share of the
// candidates with a lane-strided dot + wavefront reduction, wave 0 picks the first
// maximum (torch.argmax tie rule), the projection ratio is a second reduction and the
// residual is updated in place; nothing leaves the chip between sources.
// HBM-bound and tiny (800 workgroups x 37 x 3072 MACs at config 2).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// pred element (b, p, t, k) at pred[b*sb + p*sp + t*st + k*sk]; bank [ncand, nf2, np]; ss [nb, nt, ncand]; idx [nb, nt, nsrc]; vad [nb, nt, nsrc]
__global__ void __launch_bounds__(256)
ipd2doa_kernel(const float* __restrict__ pred, long long sb, long long sp, long long st, long long sk,
               const float* __restrict__ bank, int nb, int np, int nt, int nf2,
               int ncand, int nsrc, int unk_num, float* __restrict__ ss, int* __restrict__ idx,
               float* __restrict__ vad) {
  extern __shared__ float smem[];
  float* res = smem;                 // [nf2 * np] residual IPD vector, index k*np + p (reference flattening)
  float* score = smem + nf2 * np;    // [ncand]
  __shared__ int best_s;
  __shared__ float ratio_s;
  const int b = blockIdx.x / nt, t = blockIdx.x - b * nt;
  const int X = nf2 * np;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < X; i += 256) {
    const int k = i / np, p = i - k * np;
    res[i] = pred[b * sb + p * sp + t * st + k * sk];
  }
  __syncthreads();
  const float norm = (float)(np * nf2) * 0.5f;
  for (int s = 0; s < nsrc; ++s) {
    for (int c = wave; c < ncand; c += 4) {
      const float* tp = bank + (long long)c * X;
      float acc = 0.f;
      for (int i = lane; i < X; i += 64) acc = fmaf(res[i], tp[i], acc);
      acc = wave_sum(acc);
      if (lane == 0) score[c] = __fdiv_rn(acc, norm);
    }
    __syncthreads();
    if (s == 0)
      for (int c = tid; c < ncand; c += 256) ss[((long long)b * nt + t) * ncand + c] = score[c];
    if (wave == 0) {
      // first maximum (torch.argmax): lane-strided scan, then reduce (value desc, index asc)
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < ncand; c += 64)
        if (score[c] > bv) {
          bv = score[c];
          bi = c;
        }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(bv, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      if (lane == 0) best_s = bi;
    }
    __syncthreads();
    const int best = best_s;
    const float* tp = bank + (long long)best * X;
    if (wave == 0) {
      float num = 0.f, den = 0.f;
      for (int i = lane; i < X; i += 64) {
        const float tv = tp[i];
        num = fmaf(tv, res[i], num);
        den = fmaf(tv, tv, den);
      }
      num = wave_sum(num);
      den = wave_sum(den);
      if (lane == 0) {
        const float r = __fdiv_rn(num, den);
        ratio_s = r;
        idx[((long long)b * nt + t) * nsrc + s] = best;
        vad[((long long)b * nt + t) * nsrc + s] = unk_num ? r : 1.0f;
      }
    }
    __syncthreads();
    const float r = ratio_s;
    for (int i = tid; i < X; i += 256) res[i] = __fsub_rn(res[i], __fmul_rn(r, tp[i]));
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int fnssl_ipd2doa(const float* pred, long long sb, long long sp, long long st, long long sk, const float* bank, int nb,
                  int np, int nt, int nf2, int ncand, int nsrc, int unk_num, float* ss, int* idx, float* vad,
                  void* stream) {
  FNSSL_REQUIRE(nb > 0 && np > 0 && nt >= 0 && nf2 > 0 && ncand > 0 && nsrc > 0, "ipd2doa: bad sizes");
  if (nt == 0) return FNSSL_OK;
  FNSSL_REQUIRE(pred && bank && ss && idx && vad, "ipd2doa: null pointer");
  const size_t lds = ((size_t)nf2 * np + ncand) * sizeof(float);
  FNSSL_REQUIRE(lds <= 60 * 1024, "ipd2doa: 2nf*np = %d does not fit the LDS budget", nf2 * np);
  FNSSL_REQUIRE((long long)nb * nt < (1ll << 31), "ipd2doa: too many segments");
  fnssl::TimedLaunch tl("ipd2doa", fnssl::as_stream(stream));
  hipLaunchKernelGGL(ipd2doa_kernel, dim3(nb * nt), dim3(256), lds, fnssl::as_stream(stream), pred, sb, sp, st, sk, bank,
                     nb, np, nt, nf2, ncand, nsrc, unk_num, ss, idx, vad);
  FNSSL_CHECK_LAUNCH("ipd2doa_kernel");
  return FNSSL_OK;
}

}  // extern "C"
