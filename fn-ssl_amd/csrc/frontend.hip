// Front end of the FN-SSL DP-IPD path on gfx950: multi-channel STFT, mic-pair
// re-batching + recursive magnitude normalisation + feature packing, and the
// [n,c,f,t] -> [n,t,f,c] relayout.  All HBM-bound (28 KB per
// utterance-frame), so the design goal is coalesced traffic and single passes:
//   * stft_kernel: one wave per (utterance, channel, frame); 512 real samples are
//     packed into a 256-point complex FFT (radix-2 butterflies staged in LDS),
//     split into 257 bins in registers, written k-contiguous, and |X| is reduced
//     across the wave so the normalisation never re-reads the spectrum;
//   * ema_kernel: the 300-step recursion runs once per mic pair on the per-frame
//     sums (the reference re-launches ~5 kernels per frame, utils.py:30-44);
//   * pack_kernel: gathers the two mics of a pair, divides by (mu + eps), drops
//     the DC bin and writes float4 = [Re i, Re j, Im i, Im j] per (pair, t, f).
#include <cmath>

#include "common.h"

namespace {

constexpr int kWin = FNSSL_WIN_LEN;   // 512
constexpr int kHop = FNSSL_HOP;       // 256
constexpr int kBins = FNSSL_NBIN;     // 257
constexpr int kNF = FNSSL_NF;         // 256
constexpr int kFramesPerBlock = 4;    // one wave each

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ unsigned bitrev8(unsigned x) { return __brev(x) >> 24; }

// Orders this wave's LDS writes before its later LDS reads (a wave's LDS operations complete in order; this keeps the
// compiler from moving them across and waits for the outstanding ones) without stopping the other waves.
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// sig[b*sb + n*sn + c*sc] -> spec [nb, nch, nt, 257] (re, im), magsum [nb, nch, nt]
// CENTER = torch.stft(center=True): frame t covers samples t*hop - 256 .. t*hop + 255 of the signal extended by
// reflection (pad_mode 'reflect': x[-n] = x[n], x[ns-1+n] = x[ns-1-n]); the index is folded per sample, nothing is
// padded in memory (IPDnet2/Module.py:62).
template <bool CENTER>
__global__ void __launch_bounds__(kFramesPerBlock * 64)
stft_kernel(const float* __restrict__ sig, int nb, int nch, int nt, int ns, int hop, long long sb, long long sn,
            long long sc, float2* __restrict__ spec, float* __restrict__ magsum) {
  __shared__ float2 tw[256];                       // exp(-2*pi*i*k/512), k = 0..255
  __shared__ float2 buf[kFramesPerBlock][256];
  const int tid = threadIdx.x;
  {
    float s, c;
    sincospif(-(float)tid / 256.0f, &s, &c);       // angle = -2*pi*tid/512
    tw[tid] = make_float2(c, s);
  }
  const int wave = tid >> 6, lane = tid & 63;
  const long long frame = (long long)blockIdx.x * kFramesPerBlock + wave;   // over (b, c, t)
  const long long nframes = (long long)nb * nch * nt;
  const bool active = frame < nframes;
  const long long fr = active ? frame : nframes - 1;
  const int t = (int)(fr % nt);
  const int c = (int)((fr / nt) % nch);
  const int b = (int)(fr / ((long long)nt * nch));
  const float* chan = sig + b * sb + c * sc;
  const int n0 = t * hop - (CENTER ? kWin / 2 : 0);
  float2* z = buf[wave];

  // z[m] = w[2m] x[2m] + i w[2m+1] x[2m+1], stored bit-reversed for the DIT FFT
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = lane + 64 * r;
    int i0 = n0 + 2 * m, i1 = i0 + 1;
    if (CENTER) {
      i0 = i0 < 0 ? -i0 : i0;
      i1 = i1 < 0 ? -i1 : i1;
      i0 = i0 >= ns ? 2 * (ns - 1) - i0 : i0;
      i1 = i1 >= ns ? 2 * (ns - 1) - i1 : i1;
    }
    const float x0 = chan[(long long)i0 * sn];
    const float x1 = chan[(long long)i1 * sn];
    const float w0 = 0.5f - 0.5f * cospif((float)(2 * m) / 256.0f);       // periodic Hann-512
    const float w1 = 0.5f - 0.5f * cospif((float)(2 * m + 1) / 256.0f);
    z[bitrev8(m)] = make_float2(w0 * x0, w1 * x1);
  }
  __syncthreads();                                 // the twiddle table is shared by the block's waves
  // 8 radix-2 stages of the 256-point complex FFT; 128 butterflies per stage, 2 per lane
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int half = 1 << st;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int bf = lane + 64 * r;              // butterfly id 0..127
      const int j = bf & (half - 1);
      const int i0 = ((bf >> st) << (st + 1)) + j;
      const int i1 = i0 + half;
      const float2 w = tw[j << (8 - st)];        // exp(-2 pi i j / (2 half)) = tw[j * 512/(2 half)]
      const float2 a = z[i0], bb = z[i1];
      const float2 tb = make_float2(bb.x * w.x - bb.y * w.y, bb.x * w.y + bb.y * w.x);
      z[i0] = make_float2(a.x + tb.x, a.y + tb.y);
      z[i1] = make_float2(a.x - tb.x, a.y - tb.y);
    }
    wave_lds_sync();                               // z is private to the wave: no workgroup barrier between stages
  }
  // split: X[k] = E[k] + W512^k O[k],  E = (Z[k] + conj Z[N-k])/2,  O = (Z[k] - conj Z[N-k])/(2i)
  float2* out = spec + fr * kBins;
  float msum = 0.f;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int k = lane + 64 * r;
    if (k <= 256) {
      const float2 zk = z[k & 255];
      const float2 zn = z[(256 - k) & 255];
      const float er = 0.5f * (zk.x + zn.x), ei = 0.5f * (zk.y - zn.y);
      const float dr = 0.5f * (zk.x - zn.x), di = 0.5f * (zk.y + zn.y);   // (Z[k] - conj Z[N-k]) / 2
      const float orr = di, oi = -dr;                                        // divide by i
      float2 w = k < 256 ? tw[k] : make_float2(-1.f, 0.f);
      const float xr = er + (orr * w.x - oi * w.y);
      const float xi = ei + (orr * w.y + oi * w.x);
      if (active) out[k] = make_float2(xr, xi);
      msum += sqrtf(xr * xr + xi * xi);
    }
  }
  msum = wave_sum(msum);
  if (active && lane == 0 && magsum) magsum[fr] = msum;
}

__device__ __forceinline__ void pair_of(int p, int nch, int ch_mode, int& mi, int& mj) {
  if (ch_mode == FNSSL_CH_MODE_M) {
    mi = 0;
    mj = p + 1;
    return;
  }
  int i = 0, left = p;
  while (left >= nch - 1 - i) {   // row i holds nch-1-i pairs (Module.py:397-402)
    left -= nch - 1 - i;
    ++i;
  }
  mi = i;
  mj = i + 1 + left;
}

// mu[pair, t]: mu_t = a_t * mu_{t-1} + b_t * mean_t, mean_t = (S_i + S_j) / 514.
// One WAVE per pair: the lanes form the per-frame means and the products b_t * mean_t in parallel (coalesced reads)
// into LDS, lane 0 runs the nt-step recursion on LDS operands (a dependent chain of one multiply and one add per
// frame instead of nt round trips to memory), the lanes write mu back coalesced.  Products and the sum are rounded
// separately like the reference's tensor ops (utils.py:33-41); frames beyond kEmaTile continue from the carried mu.
constexpr int kEmaTile = 1024;
constexpr int kEmaWaves = 4;

__device__ __forceinline__ void ema_scan_tile(float* bm, const float* ca_t, int n, float& m, int lane) {
  // bm[0..n) holds b_t * mean_t on entry and mu_t on exit; ca_t[0..n) the a_t of the same frames
  wave_lds_sync();
  if (lane == 0) {
    float mm = m;
    for (int t = 0; t < n; ++t) {
      mm = __fadd_rn(__fmul_rn(ca_t[t], mm), bm[t]);
      bm[t] = mm;
    }
    m = mm;
  }
  wave_lds_sync();
}

__global__ void __launch_bounds__(kEmaWaves * 64)
ema_kernel(const float* __restrict__ magsum, const float* __restrict__ ca, const float* __restrict__ cb, int nb,
           int nch, int np, int nt, int ch_mode, float* __restrict__ mu) {
  __shared__ float bm_s[kEmaWaves][kEmaTile], ca_s[kEmaWaves][kEmaTile];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * kEmaWaves + wave;
  if (idx >= nb * np) return;
  const int b = idx / np, p = idx - b * np;
  int mi, mj;
  pair_of(p, nch, ch_mode, mi, mj);
  const float* si = magsum + ((long long)b * nch + mi) * nt;
  const float* sj = magsum + ((long long)b * nch + mj) * nt;
  float m = 0.f;
  for (int t0 = 0; t0 < nt; t0 += kEmaTile) {
    const int n = min(kEmaTile, nt - t0);
    for (int t = lane; t < n; t += 64) {
      const float mean = __fdiv_rn(__fadd_rn(si[t0 + t], sj[t0 + t]), (float)(2 * kBins));
      bm_s[wave][t] = __fmul_rn(cb[t0 + t], mean);
      ca_s[wave][t] = ca[t0 + t];
    }
    ema_scan_tile(bm_s[wave], ca_s[wave], n, m, lane);
    for (int t = lane; t < n; t += 64) mu[(long long)idx * nt + t0 + t] = bm_s[wave][t];
    __builtin_amdgcn_wave_barrier();
  }
}

// x[pair, t, f, 0..3] = [Re i, Re j, Im i, Im j](bin f+1) / (mu + eps)
template <int LAYOUT>
__global__ void __launch_bounds__(256)
pack_kernel(const float2* __restrict__ spec, const float* __restrict__ mu, int nb, int nch, int np,
            int nt, int ch_mode, float eps, float* __restrict__ x) {
  const long long row = blockIdx.x;   // (pair index, t)
  const int f = threadIdx.x;          // 0..255  <-> bin f+1
  const int t = (int)(row % nt);
  const long long idx = row / nt;
  const int b = (int)(idx / np), p = (int)(idx - (long long)b * np);
  int mi, mj;
  pair_of(p, nch, ch_mode, mi, mj);
  const float2 xi = spec[(((long long)b * nch + mi) * nt + t) * kBins + f + 1];
  const float2 xj = spec[(((long long)b * nch + mj) * nt + t) * kBins + f + 1];
  const float den = __fadd_rn(mu[idx * nt + t], eps);
  const float4 o = make_float4(__fdiv_rn(xi.x, den), __fdiv_rn(xj.x, den), __fdiv_rn(xi.y, den),
                               __fdiv_rn(xj.y, den));
  if (LAYOUT == 0) {
    reinterpret_cast<float4*>(x)[(idx * nt + t) * kNF + f] = o;
  } else {
    float* base = x + idx * 4 * (long long)kNF * nt + (long long)f * nt + t;
    const long long cs = (long long)kNF * nt;
    base[0] = o.x;
    base[cs] = o.y;
    base[2 * cs] = o.z;
    base[3 * cs] = o.w;
  }
}

// x [n, c, nf, nt] -> y [n, nt, nf, c]; 32x32 (f, t) tiles through LDS so both sides coalesce
__global__ void __launch_bounds__(256)
nchw_to_seq_kernel(const float* __restrict__ x, int c, int nf, int nt, float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z / c, ch = blockIdx.z - n * c;
  const int f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* src = x + ((long long)n * c + ch) * nf * nt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + ty + 8 * r, t = t0 + tx;
    if (f < nf && t < nt) tile[ty + 8 * r][tx] = src[(long long)f * nt + t];
  }
  __syncthreads();
  float* dst = y + (long long)n * nt * nf * c + ch;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = t0 + ty + 8 * r, f = f0 + tx;
    if (f < nf && t < nt) dst[((long long)t * nf + f) * c] = tile[tx][ty + 8 * r];
  }
}

// ---- all-channel ("array") features of IPDnet (reference IPDnet/runIPDnetOn.py:240-254) ----------
// mu[b, t]: the same recursion on the mean magnitude over ALL channels and the 257 bins (one wave per utterance,
// like ema_kernel; the channel sum keeps the reference's left-to-right order).
__global__ void __launch_bounds__(kEmaWaves * 64)
ema_array_kernel(const float* __restrict__ magsum, const float* __restrict__ ca, const float* __restrict__ cb, int nb,
                 int nch, int nt, float* __restrict__ mu) {
  __shared__ float bm_s[kEmaWaves][kEmaTile], ca_s[kEmaWaves][kEmaTile];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * kEmaWaves + wave;
  if (b >= nb) return;
  const float* s = magsum + (long long)b * nch * nt;
  const float cnt = (float)(nch * kBins);
  float m = 0.f;
  for (int t0 = 0; t0 < nt; t0 += kEmaTile) {
    const int n = min(kEmaTile, nt - t0);
    for (int t = lane; t < n; t += 64) {
      float acc = s[t0 + t];
      for (int c = 1; c < nch; ++c) acc = __fadd_rn(acc, s[(long long)c * nt + t0 + t]);
      const float mean = __fdiv_rn(acc, cnt);
      bm_s[wave][t] = __fmul_rn(cb[t0 + t], mean);
      ca_s[wave][t] = ca[t0 + t];
    }
    ema_scan_tile(bm_s[wave], ca_s[wave], n, m, lane);
    for (int t = lane; t < n; t += 64) mu[(long long)b * nt + t0 + t] = bm_s[wave][t];
    __builtin_amdgcn_wave_barrier();
  }
}

// x[b, t, f, :] = [Re ch 0..nch-1, Im ch 0..nch-1](bin f+1) / (mu + eps)
template <int LAYOUT>
__global__ void __launch_bounds__(256)
pack_array_kernel(const float2* __restrict__ spec, const float* __restrict__ mu, int nch, int nt, float eps,
                  float* __restrict__ x) {
  extern __shared__ float row_lds[];   // LAYOUT 0: the (b, t) row [256 bins][2 nch] = one contiguous piece of x
  const long long row = blockIdx.x;   // (b, t)
  const int f = threadIdx.x;
  const int t = (int)(row % nt);
  const long long b = row / nt;
  const float den = __fadd_rn(mu[row], eps);
  for (int c = 0; c < nch; ++c) {
    const float2 v = spec[((b * nch + c) * nt + t) * kBins + f + 1];
    const float re = __fdiv_rn(v.x, den), im = __fdiv_rn(v.y, den);
    if (LAYOUT == 0) {
      // staged: bin-major 4-byte stores straight to x were 2 nch partial writes of a 64-byte stride per lane
      row_lds[f * (2 * nch) + c] = re;
      row_lds[f * (2 * nch) + nch + c] = im;
    } else {
      float* o = x + b * 2 * nch * (long long)kNF * nt + (long long)f * nt + t;
      const long long cs = (long long)kNF * nt;
      o[c * cs] = re;
      o[(nch + c) * cs] = im;
    }
  }
  if (LAYOUT == 0) {
    __syncthreads();
    float* o = x + row * kNF * (2 * nch);
    const int n = kNF * 2 * nch;                       // multiple of 4; the row starts on a 16-byte boundary
    for (int i = 4 * f; i < n; i += 4 * 256)
      *reinterpret_cast<float4*>(o + i) = *reinterpret_cast<const float4*>(row_lds + i);
  }
}

// layout 1 ([nb, 2 nch, 256, nt], frames fastest — the tensor the reference builds) as a tiled transposition: a
// workgroup reads a 64-frame x 32-bin tile of one channel's spectrum bin-contiguous (256-byte runs), normalises it and
// writes both planes (Re -> channel c, Im -> channel nch + c) frame-contiguous (256-byte runs).  The element-wise
// version (pack_array_kernel<1>) wrote one float per nt*4-byte stride.
__global__ void __launch_bounds__(256)
pack_array_planes_kernel(const float2* __restrict__ spec, const float* __restrict__ mu, int nch, int nt, float eps,
                         float* __restrict__ x) {
  __shared__ float re_t[32][65], im_t[32][65];
  const int t0 = blockIdx.x * 64, f0 = blockIdx.y * 32;
  const long long b = blockIdx.z / nch;
  const int c = blockIdx.z - (int)b * nch;
  const int tid = threadIdx.x;
  {
    const int fl = tid & 31, tl = tid >> 5;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int tt = tl + 8 * r, t = t0 + tt;
      if (t < nt) {
        const float den = __fadd_rn(mu[b * nt + t], eps);
        const float2 v = spec[((b * nch + c) * nt + t) * kBins + f0 + fl + 1];
        re_t[fl][tt] = __fdiv_rn(v.x, den);
        im_t[fl][tt] = __fdiv_rn(v.y, den);
      }
    }
  }
  __syncthreads();
  {
    const int tt = tid & 63, fl = tid >> 6, t = t0 + tt;
    if (t < nt) {
      const long long cs = (long long)kNF * nt;
      float* o = x + b * 2 * nch * cs + (long long)f0 * nt + t;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = fl + 4 * r;
        o[c * cs + (long long)f * nt] = re_t[f][tt];
        o[(nch + c) * cs + (long long)f * nt] = im_t[f][tt];
      }
    }
  }
}

}  // namespace

extern "C" {

int fnssl_num_frames(int ns) { return ns < kWin ? 0 : (ns - kWin) / kHop + 1; }

int fnssl_num_frames_ex(int ns, int hop, int center) {
  if (hop <= 0) return 0;
  if (center) return ns > kWin / 2 ? ns / hop + 1 : 0;      // reflect padding needs ns > 256 (torch.stft's own check)
  return ns < kWin ? 0 : (ns - kWin) / hop + 1;
}

int fnssl_num_pairs(int nch, int ch_mode) {
  if (nch < 2) return 0;
  return ch_mode == FNSSL_CH_MODE_MM ? nch * (nch - 1) / 2 : nch - 1;
}

int fnssl_forgetting_coefs(int nt, int sample_length, float* a, float* b) {
  FNSSL_REQUIRE(nt >= 0 && sample_length > 0 && a && b, "forgetting_coefs: bad arguments");
  const double alpha = (double)(sample_length - 1) / (double)(sample_length + 1);
  for (int t = 0; t < nt; ++t) {
    if (t < sample_length) {
      // alp = torch.min(torch.tensor([(t-1)/(t+1), alpha])) is a float32 tensor; 1 - alp is formed in float32
      const double r = (double)(t - 1) / (double)(t + 1);
      const float alp = (float)(r < alpha ? r : alpha);
      a[t] = alp;
      b[t] = 1.0f - alp;
    } else {
      // Python doubles alpha and (1 - alpha), each rounded to float32 when it meets the tensor
      a[t] = (float)alpha;
      b[t] = (float)(1.0 - alpha);
    }
  }
  return FNSSL_OK;
}

int fnssl_stft_ex(const float* sig, int nb, int ns, int nch, long long sb, long long sn, long long sc, int hop,
                  int center, float* spec, float* magsum, void* stream) {
  FNSSL_REQUIRE(nb > 0 && nch > 0, "stft: empty batch (nb %d, nch %d)", nb, nch);
  FNSSL_REQUIRE(hop > 0 && hop <= kWin, "stft: hop %d outside 1..%d", hop, kWin);
  const int nt = fnssl_num_frames_ex(ns, hop, center);
  FNSSL_REQUIRE(nt > 0, center ? "stft: signal of %d samples is too short for reflect padding of %d"
                               : "stft: signal of %d samples is shorter than one %d-sample window",
                ns, center ? kWin / 2 : kWin);
  FNSSL_REQUIRE(sig && spec, "stft: null pointer");
  const long long nframes = (long long)nb * nch * nt;
  const long long nblk = (nframes + kFramesPerBlock - 1) / kFramesPerBlock;
  FNSSL_REQUIRE(nblk < (1ll << 31), "stft: too many frames");
  fnssl::TimedLaunch tl("stft", fnssl::as_stream(stream));
  if (center)
    hipLaunchKernelGGL(stft_kernel<true>, dim3((unsigned)nblk), dim3(kFramesPerBlock * 64), 0, fnssl::as_stream(stream),
                       sig, nb, nch, nt, ns, hop, sb, sn, sc, reinterpret_cast<float2*>(spec), magsum);
  else
    hipLaunchKernelGGL(stft_kernel<false>, dim3((unsigned)nblk), dim3(kFramesPerBlock * 64), 0, fnssl::as_stream(stream),
                       sig, nb, nch, nt, ns, hop, sb, sn, sc, reinterpret_cast<float2*>(spec), magsum);
  FNSSL_CHECK_LAUNCH("stft_kernel");
  return FNSSL_OK;
}

int fnssl_stft(const float* sig, int nb, int ns, int nch, long long sb, long long sn, long long sc, float* spec,
               float* magsum, void* stream) {
  return fnssl_stft_ex(sig, nb, ns, nch, sb, sn, sc, kHop, 0, spec, magsum, stream);
}

int fnssl_pair_features(const float* spec, const float* magsum, const float* coef_a, const float* coef_b,
                        int nb, int nch, int nt, int ch_mode, float eps, float* mu, float* x, int layout,
                        void* stream) {
  FNSSL_REQUIRE(spec && magsum && coef_a && coef_b && mu && x, "pair_features: null pointer");
  FNSSL_REQUIRE(ch_mode == FNSSL_CH_MODE_M || ch_mode == FNSSL_CH_MODE_MM, "pair_features: ch_mode %d", ch_mode);
  FNSSL_REQUIRE(layout == 0 || layout == 1, "pair_features: layout %d", layout);
  const int np = fnssl_num_pairs(nch, ch_mode);
  FNSSL_REQUIRE(nb > 0 && nt > 0 && np > 0, "pair_features: needs >= 2 channels and a non-empty batch");
  hipStream_t st = fnssl::as_stream(stream);
  {
    fnssl::TimedLaunch tl("ema", st);
    const int n = nb * np;
    hipLaunchKernelGGL(ema_kernel, dim3((n + kEmaWaves - 1) / kEmaWaves), dim3(kEmaWaves * 64), 0, st, magsum, coef_a,
                       coef_b, nb, nch, np, nt, ch_mode, mu);
    FNSSL_CHECK_LAUNCH("ema_kernel");
  }
  {
    fnssl::TimedLaunch tl("pack", st);
    const long long rows = (long long)nb * np * nt;
    FNSSL_REQUIRE(rows < (1ll << 31), "pair_features: too many rows");
    if (layout == 0)
      hipLaunchKernelGGL(pack_kernel<0>, dim3((unsigned)rows), dim3(256), 0, st,
                         reinterpret_cast<const float2*>(spec), mu, nb, nch, np, nt, ch_mode, eps, x);
    else
      hipLaunchKernelGGL(pack_kernel<1>, dim3((unsigned)rows), dim3(256), 0, st,
                         reinterpret_cast<const float2*>(spec), mu, nb, nch, np, nt, ch_mode, eps, x);
    FNSSL_CHECK_LAUNCH("pack_kernel");
  }
  return FNSSL_OK;
}

int fnssl_array_features(const float* spec, const float* magsum, const float* coef_a, const float* coef_b,
                         int nb, int nch, int nt, float eps, float* mu, float* x, int layout, void* stream) {
  FNSSL_REQUIRE(spec && magsum && coef_a && coef_b && mu && x, "array_features: null pointer");
  FNSSL_REQUIRE(layout == 0 || layout == 1, "array_features: layout %d", layout);
  FNSSL_REQUIRE(nb > 0 && nt > 0 && nch > 0, "array_features: empty problem");
  hipStream_t st = fnssl::as_stream(stream);
  {
    fnssl::TimedLaunch tl("ema", st);
    hipLaunchKernelGGL(ema_array_kernel, dim3((nb + kEmaWaves - 1) / kEmaWaves), dim3(kEmaWaves * 64), 0, st, magsum,
                       coef_a, coef_b, nb, nch, nt, mu);
    FNSSL_CHECK_LAUNCH("ema_array_kernel");
  }
  {
    fnssl::TimedLaunch tl("pack", st);
    const long long rows = (long long)nb * nt;
    FNSSL_REQUIRE(rows < (1ll << 31), "array_features: too many rows");
    {
      const size_t lds = (size_t)kNF * 2 * nch * sizeof(float);
      FNSSL_REQUIRE(lds <= 160 * 1024, "array_features: %d channels do not fit the staging row", nch);
      if (lds > 48 * 1024)
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pack_array_kernel<0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (layout == 0)
      hipLaunchKernelGGL(pack_array_kernel<0>, dim3((unsigned)rows), dim3(256), (size_t)kNF * 2 * nch * sizeof(float), st,
                         reinterpret_cast<const float2*>(spec), mu, nch, nt, eps, x);
    else {
      FNSSL_REQUIRE((long long)nb * nch < 65536, "array_features: nb * nch too large for one launch");
      hipLaunchKernelGGL(pack_array_planes_kernel, dim3((nt + 63) / 64, kNF / 32, nb * nch), dim3(256), 0, st,
                         reinterpret_cast<const float2*>(spec), mu, nch, nt, eps, x);
    }
    FNSSL_CHECK_LAUNCH("pack_array_kernel");
  }
  return FNSSL_OK;
}

int fnssl_nchw_to_seq(const float* x, int n, int c, int nf, int nt, float* y, void* stream) {
  FNSSL_REQUIRE(x && y && n > 0 && c > 0 && nf > 0 && nt > 0, "nchw_to_seq: bad arguments");
  FNSSL_REQUIRE((long long)n * c < 65536, "nchw_to_seq: n*c too large for one launch");
  fnssl::TimedLaunch tl("nchw_to_seq", fnssl::as_stream(stream));
  hipLaunchKernelGGL(nchw_to_seq_kernel, dim3((nt + 31) / 32, (nf + 31) / 32, n * c), dim3(256), 0,
                     fnssl::as_stream(stream), x, c, nf, nt, y);
  FNSSL_CHECK_LAUNCH("nchw_to_seq_kernel");
  return FNSSL_OK;
}

}  // extern "C"
