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

// ---------------------------------------------------------------------------------------------------------------------
// DP-IPD TARGETS of the training step (reference: DPIPD.forward(source_doa), FN-SSL/Lightning/Module.py:464-498, and the
// ground-truth half of MyModel.data_preprocess, FN-SSL/Lightning/main.py:227-262): for every (utterance, segment, pair,
// bin) the direct-path phase difference of each source, exp(+j 2 pi f tau) with tau = r(doa) . (mic_i - mic_j) / c
// (pair (i, j), i < j: the reference's data_adjust keeps [m1 = i, m2 = j] of its [nmic x nmic] table), masked by the
// source's voice activity (mean over the segment's VAD frames > 0) and summed over the sources:
//     ipd[b, s, k, p] = sum_src vad * cos(phase)      k <  nf_used
//     ipd[b, s, nf_used + k, p] = sum_src vad * sin(phase)
// The reference forms the phase in float64 (numpy) and rounds the cos / sin to float32: so does this kernel (a few
// thousand double-precision sincos per utterance — nothing to optimise).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256)
dpipd_targets_kernel(const float* __restrict__ doa, const float* __restrict__ vad, int nb, int nseg, int nvad, int ns,
                     const float* __restrict__ mic, int nmic, int ch_mode, int np, int bin0, int nf_used, int nbins,
                     float fre_max, float speed, int use_vad, float* __restrict__ ipd, float* __restrict__ vad_mean) {
  const int bs = blockIdx.x;                  // (utterance, segment)
  __shared__ double tau_s[64 * 8];            // [source][pair], <= 8 sources x 64 pairs
  __shared__ float gate_s[8];
  const int tid = threadIdx.x;
  if (tid < ns) {
    float m = 0.f;
    for (int v = 0; v < nvad; ++v) m += vad[((size_t)bs * nvad + v) * ns + tid];     // vad [nb, nseg, nvad, ns]
    m = nvad > 0 ? m / (float)nvad : 1.f;                                             // vad_batch.mean(axis=2)
    if (vad_mean) vad_mean[(size_t)bs * ns + tid] = m;
    gate_s[tid] = use_vad ? (m > 0.f ? 1.f : 0.f) : 1.f;                              // th = 0 (main.py:251-253)
  }
  for (int i = tid; i < ns * np; i += 256) {
    const int src = i / np, p = i - src * np;
    int mi = 0, mj = p + 1;                                                           // 'M': pairs (0, j)
    if (ch_mode == FNSSL_CH_MODE_MM) {                                                // 'MM': pairs (i, j), i < j, i-major
      int rem = p;
      mi = 0;
      while (rem >= nmic - 1 - mi) {
        rem -= nmic - 1 - mi;
        ++mi;
      }
      mj = mi + 1 + rem;
    }
    const double ele = doa[((size_t)bs * 2 + 0) * ns + src], azi = doa[((size_t)bs * 2 + 1) * ns + src];   // doa [nb, nseg, 2, ns]
    const double rx = sin(ele) * cos(azi), ry = sin(ele) * sin(azi), rz = cos(ele);
    const double dx = (double)mic[mi * 3 + 0] - (double)mic[mj * 3 + 0], dy = (double)mic[mi * 3 + 1] - (double)mic[mj * 3 + 1],
                 dz = (double)mic[mi * 3 + 2] - (double)mic[mj * 3 + 2];
    tau_s[src * np + p] = (rx * dx + ry * dy + rz * dz) / (double)speed;              // ITD[m1 = i, m2 = j] (Module.py:488)
  }
  __syncthreads();
  const double two_pi = 6.283185307179586476925286766559;
  for (int i = tid; i < nf_used * np; i += 256) {
    const int k = i / np, p = i - k * np;
    const double f = (double)(bin0 + k) * ((double)fre_max / (double)(nbins - 1));    // np.linspace(0, fre_max, nf)[bin0 + k] = k * step
    float re = 0.f, im = 0.f;
    for (int src = 0; src < ns; ++src) {
      const double ph = two_pi * f * tau_s[src * np + p];                             // (-2 pi f ITD) * (-1), Module.py:489-490
      re += gate_s[src] * (float)cos(ph);                                             // float32 sum over sources (torch.sum, main.py:258)
      im += gate_s[src] * (float)sin(ph);
    }
    float* o = ipd + (size_t)bs * (2 * nf_used) * np;
    o[(size_t)k * np + p] = re;
    o[(size_t)(nf_used + k) * np + p] = im;
  }
}

}  // namespace

extern "C" int fnssl_dpipd_targets(const float* doa, const float* vad, int nb, int nseg, int nvad, int ns, const float* mic_loc,
                                   int nmic, int ch_mode, int bin0, int nf_used, int nbins, float fre_max, float speed,
                                   int use_vad, float* ipd, float* vad_mean, void* stream) {
  FNSSL_REQUIRE(doa && mic_loc && ipd && (vad || nvad == 0), "dpipd_targets: null pointer");
  FNSSL_REQUIRE(nb > 0 && nseg > 0 && ns >= 1 && ns <= 8 && nmic >= 2 && nvad >= 0, "dpipd_targets: nb %d nseg %d sources %d (1..8) mics %d", nb,
                nseg, ns, nmic);
  FNSSL_REQUIRE(ch_mode == FNSSL_CH_MODE_M || ch_mode == FNSSL_CH_MODE_MM, "dpipd_targets: channel mode %d", ch_mode);
  const int np = fnssl_num_pairs(nmic, ch_mode);
  FNSSL_REQUIRE(np >= 1 && np <= 64, "dpipd_targets: %d microphone pairs (at most 64)", np);
  FNSSL_REQUIRE(nbins >= 2 && bin0 >= 0 && nf_used >= 1 && bin0 + nf_used <= nbins && fre_max > 0.f && speed > 0.f,
                "dpipd_targets: bins [%d, %d) of %d", bin0, bin0 + nf_used, nbins);
  hipLaunchKernelGGL(dpipd_targets_kernel, dim3(nb * nseg), dim3(256), 0, fnssl::as_stream(stream), doa, vad, nb, nseg, nvad, ns,
                     mic_loc, nmic, ch_mode, np, bin0, nf_used, nbins, fre_max, speed, use_vad, ipd, vad_mean);
  FNSSL_CHECK_LAUNCH("dpipd_targets_kernel");
  return FNSSL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Peak detection over the spatial spectrum (reference: SourceDetectLocalize.forward, meth_mode 'PD',
// FN-SSL/Lightning/Module.py:580-611): the last azimuth column is dropped as redundant (:581); a cell is a peak when it is
// strictly larger than its 8 neighbours — azimuth circular over the remaining columns, elevation clamped (:583-598; the clamp
// compares the first and last elevation row with themselves, so they never hold a peak) —; per frame the peaks are sorted by
// value, descending, equal values in ascending flat-index order (python's stable sorted(), :608-609) and the first nsrc kept.
// One workgroup per frame: the spectrum (<= 12 K cells) in LDS, every thread marks its cells, then nsrc rounds of a
// block-wide (value desc, index asc) argmax.  The reference does this in a Python double loop with a sort per frame.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256)
doa_peaks_kernel(const float* __restrict__ ss, int nele, int nazi, int nsrc, int* __restrict__ idx, float* __restrict__ val,
                 int* __restrict__ count) {
  extern __shared__ float smem[];
  float* g = smem;                          // [nele * nazi]
  float* pk = smem + nele * nazi;           // [nele * nazi]: the cell's value if it is a peak, else -inf
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ int npk_s;
  const int ncell = nele * nazi, w = nazi - 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = ss + (size_t)blockIdx.x * ncell;
  if (tid == 0) npk_s = 0;
  for (int i = tid; i < ncell; i += 256) g[i] = src[i];
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < ncell; i += 256) {
    const int e = i / nazi, a = i - e * nazi;
    bool ok = a < w;
    if (ok) {
      const float v = g[i];
      const int e0 = e > 0 ? e - 1 : 0, e1 = e < nele - 1 ? e + 1 : nele - 1;
      const int a0 = a > 0 ? a - 1 : w - 1, a1 = a < w - 1 ? a + 1 : 0;
      ok = v > g[e0 * nazi + a] && v > g[e1 * nazi + a] && v > g[e * nazi + a0] && v > g[e * nazi + a1] &&
           v > g[e0 * nazi + a0] && v > g[e0 * nazi + a1] && v > g[e1 * nazi + a0] && v > g[e1 * nazi + a1];
    }
    pk[i] = ok ? g[i] : -INFINITY;
    mine += ok ? 1 : 0;
  }
  if (mine) atomicAdd(&npk_s, mine);
  __syncthreads();
  const int npk = npk_s;
  if (tid == 0) count[blockIdx.x] = npk < nsrc ? npk : nsrc;
  for (int s = 0; s < nsrc; ++s) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (s < npk)
      for (int i = tid; i < ncell; i += 256)
        if (pk[i] > bv) {                  // ascending i per thread: the first of equal values stays
          bv = pk[i];
          bi = i;
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
    if (lane == 0) {
      wv[wave] = bv;
      wi[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < 4; ++k)
        if (wv[k] > bv || (wv[k] == bv && wi[k] < bi)) {
          bv = wv[k];
          bi = wi[k];
        }
      const bool have = s < npk;
      idx[(size_t)blockIdx.x * nsrc + s] = have ? bi : -1;
      val[(size_t)blockIdx.x * nsrc + s] = have ? bv : 0.f;
      if (have) pk[bi] = -INFINITY;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int fnssl_doa_peaks(const float* ss, int nframes, int nele, int nazi, int nsrc, int* idx, float* val, int* count,
                               void* stream) {
  FNSSL_REQUIRE(nframes >= 0 && nele >= 1 && nazi >= 2 && nsrc >= 1 && nsrc <= 8, "doa_peaks: %d frames, grid %d x %d, %d sources (1..8)",
                nframes, nele, nazi, nsrc);
  if (nframes == 0) return FNSSL_OK;
  FNSSL_REQUIRE(ss && idx && val && count, "doa_peaks: null pointer");
  const size_t lds = (size_t)nele * nazi * 2 * sizeof(float);
  FNSSL_REQUIRE(lds <= 96 * 1024, "doa_peaks: a %d x %d grid does not fit the LDS budget", nele, nazi);
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(doa_peaks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fnssl::TimedLaunch tl("doa_peaks", fnssl::as_stream(stream));
  hipLaunchKernelGGL(doa_peaks_kernel, dim3(nframes), dim3(256), lds, fnssl::as_stream(stream), ss, nele, nazi, nsrc, idx, val, count);
  FNSSL_CHECK_LAUNCH("doa_peaks_kernel");
  return FNSSL_OK;
}
