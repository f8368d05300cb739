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
#include <cstdlib>

#include "common.h"
#include "tuning.h"

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

// ---- frame-row kernels (round 3) ------------------------------------------------------------------------------
// One workgroup per (utterance, frame), its four waves take the channels in turn.
//   * The frame's 512 samples of ALL channels are fetched once, cooperatively and coalesced for either waveform layout
//     ([nb, ns, nch]: one contiguous 512 x nch piece; [nb, nch, ns]: nch contiguous 2-KB pieces), into an LDS image
//     [channel][513] (the odd stride keeps the interleaved-layout writes conflict-free); the centred framing of
//     torch.stft (reflection at both ends) is an index fold in that fetch.  stft_kernel read them straight from memory,
//     per wave, at a 4 * nch-byte stride.
//   * The 256-point complex FFT of the packed real frame runs as FOUR radix-4 stages in the wave's LDS buffer (one
//     butterfly per lane and stage, digit-reversed input) instead of eight radix-2 stages: half the LDS round trips.
//   * MODE 1 (magnitudes) writes only sum_k |X[k]| per (utterance, channel, frame); MODE 2 (features) runs the transform
//     AGAIN, divides by (mu + eps) and stores the frame's feature row [256 bins][Re ch 0.. | Im ch 0..] as ONE contiguous
//     piece.  The array front end (IPDnet, IPDnet2) is MODE 1 -> recursive mean -> MODE 2: the 2 x 257 x 8 B per channel-
//     frame spectrum is never written or re-read (1.1 GB of traffic instead of 1.8 GB at BASELINE config 5); MODE 0 is
//     the pair path's spectrum + magnitude-sum pass.
constexpr int kXsStride = kWin + 1;

__device__ __forceinline__ unsigned digitrev4_8(unsigned m) {      // reverse the four base-4 digits of an 8-bit index
  return ((m & 3u) << 6) | ((m & 12u) << 2) | ((m & 48u) >> 2) | ((m & 192u) >> 6);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

// 256-point complex FFT in place in z[256] (digit-reversed on entry, natural order on exit); tw[k] = exp(-2 pi i k / 512)
__device__ __forceinline__ void fft256_radix4(float2* z, const float2* tw, int lane) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int L = 1 << (2 * s);
    const int j = lane & (L - 1);
    const int base = ((lane >> (2 * s)) << (2 * s + 2)) + j;
    const int step = j * (128 >> (2 * s));                    // W_{4L}^j = tw[j * 512 / (4 L)]
    const float2 a0 = z[base];
    float2 a1 = z[base + L], a2 = z[base + 2 * L], a3 = z[base + 3 * L];
    if (s > 0) {
      a1 = cmul(a1, tw[step]);
      a2 = cmul(a2, tw[2 * step]);
      a3 = cmul(a3, tw[3 * step]);
    }
    const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y), t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
    const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
    const float2 t3 = make_float2(a1.y - a3.y, a3.x - a1.x);  // -i (a1 - a3)
    z[base] = make_float2(t0.x + t2.x, t0.y + t2.y);
    z[base + L] = make_float2(t1.x + t3.x, t1.y + t3.y);
    z[base + 2 * L] = make_float2(t0.x - t2.x, t0.y - t2.y);
    z[base + 3 * L] = make_float2(t1.x - t3.x, t1.y - t3.y);   // a butterfly is in place: only the NEXT stage reads across lanes
    wave_lds_sync();
  }
}

struct RowParams {
  const float* sig;
  int nb, nch, nt, ns, hop, center;
  long long sb, sn, sc;
  float2* spec;          // MODE 0: [nb, nch, nt, 257]
  float* magsum;         // MODE 0 / 1: [nb, nch, nt]
  const float* mu;       // MODE 2: [nb, nt]
  float eps;
  float* x;              // MODE 2: [nb, nt, 256, 2 nch]
};

// NW waves per workgroup: as many as there are channels (rounded up to 4, 8 or 16), so that a frame's channels are
// transformed side by side.  Measured (profiles/r03/g_*, BASELINE config 5: 240 k channel-frames): 0.41 ms per pass
// against 0.55 ms for stft_kernel; what bounds it is the transform itself — ~530 vector instructions and ~70 LDS
// operations (several with 4- to 8-way bank conflicts: the digit-reversed store, the stride-4 / stride-16 stages) per
// channel-frame, ≈ 0.2 ms of pure VALU issue per pass — not the memory traffic, so the spectrum-free two-pass front
// end (fnssl_array_frontend: two transform passes) is SLOWER than writing the spectrum once (0.92 vs 0.74 ms).
template <int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) stft_rows_kernel(const RowParams p) {
  constexpr int kRowWaves = NW;
  extern __shared__ __attribute__((aligned(16))) float smem_rows[];
  float2* tw = reinterpret_cast<float2*>(smem_rows);                    // [384]
  float* hann = smem_rows + 768;                                          // [512]
  float2* zbuf = reinterpret_cast<float2*>(smem_rows + 768 + 512);        // [4][256]
  float* xs = smem_rows + 768 + 512 + kRowWaves * 512;                    // [nch][513]
  float* row = xs + p.nch * kXsStride;                                    // MODE 2: [256][2 nch]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < 384; k += kRowWaves * 64) {
    float sn_, cs_;
    sincospif(-(float)k / 256.0f, &sn_, &cs_);
    tw[k] = make_float2(cs_, sn_);
  }
  for (int n = tid; n < kWin; n += kRowWaves * 64) hann[n] = 0.5f - 0.5f * cospif((float)n / 256.0f);
  // Persistent workgroup: it walks frames blockIdx.x, + gridDim.x, ...; the NEXT frame's samples are requested into
  // registers before the current frame is transformed (a workgroup per frame left the ~2 us of HBM latency of its own
  // fetch exposed: one workgroup fits a CU) and are written to the LDS image after it.
  const long long nframes = (long long)p.nb * p.nt;
  const int total = p.nch * kWin;
  constexpr int PF = 8;                             // samples per thread: nch * 512 <= NW * 64 * 8 by the choice of NW
  float pre[PF];
  const float inv = 1.0f / (float)p.nch;
  auto fetch = [&](long long fr) {
    const int t = (int)(fr % p.nt);
    const long long b = fr / p.nt;
    const int n0 = t * p.hop - (p.center ? kWin / 2 : 0);
    const float* ub = p.sig + b * p.sb;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int i = tid + q * kRowWaves * 64;
      int n, c;
      if (p.sn == 1) {             // [nb, nch, ns]: channel-major, 2-KB runs
        c = i >> 9;
        n = i & (kWin - 1);
      } else {                     // [nb, ns, nch] (sc == 1: one contiguous piece) or any other strides
        n = (int)(((float)i + 0.5f) * inv);          // i / nch without the integer division (exact: i < 2^16)
        c = i - n * p.nch;
        if (c < 0) { c += p.nch; --n; }
        if (c >= p.nch) { c -= p.nch; ++n; }
      }
      int src = n0 + n;
      if (p.center) {              // torch.stft(center=True): reflection at both ends as an index fold
        src = src < 0 ? -src : src;
        src = src >= p.ns ? 2 * (p.ns - 1) - src : src;
      }
      pre[q] = i < total ? ub[(long long)src * p.sn + (long long)c * p.sc] : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int i = tid + q * kRowWaves * 64;
      if (i < total) {
        int n, c;
        if (p.sn == 1) {
          c = i >> 9;
          n = i & (kWin - 1);
        } else {
          n = (int)(((float)i + 0.5f) * inv);
          c = i - n * p.nch;
          if (c < 0) { c += p.nch; --n; }
          if (c >= p.nch) { c -= p.nch; ++n; }
        }
        xs[c * kXsStride + n] = pre[q];
      }
    }
  };
  long long fr = blockIdx.x;
  if (fr < nframes) fetch(fr);
  float2* z = zbuf + wave * 256;
  for (; fr < nframes; fr += gridDim.x) {
    __syncthreads();               // the previous frame's transforms (and row store) are done with xs / row; tables ready
    commit();
    __syncthreads();
    if (fr + gridDim.x < nframes) fetch(fr + gridDim.x);
    const int t = (int)(fr % p.nt);
    const long long b = fr / p.nt;
    float den = 1.f;
    if (MODE == 2) den = __fadd_rn(p.mu[b * p.nt + t], p.eps);
    for (int c = wave; c < p.nch; c += kRowWaves) {
      const float* xc = xs + c * kXsStride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = lane + 64 * r;
        z[digitrev4_8(m)] = make_float2(hann[2 * m] * xc[2 * m], hann[2 * m + 1] * xc[2 * m + 1]);
      }
      wave_lds_sync();
      fft256_radix4(z, tw, lane);
      // split: X[k] = E[k] + W512^k O[k],  E = (Z[k] + conj Z[N-k])/2,  O = (Z[k] - conj Z[N-k])/(2i)
      const long long cf = (b * p.nch + c) * p.nt + t;
      float msum = 0.f;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int k = lane + 64 * r;
        if (k <= 256) {
          const float2 zk = z[k & 255];
          const float2 zn = z[(256 - k) & 255];
          const float er = 0.5f * (zk.x + zn.x), ei = 0.5f * (zk.y - zn.y);
          const float dr = 0.5f * (zk.x - zn.x), di = 0.5f * (zk.y + zn.y);
          const float orr = di, oi = -dr;
          const float2 w = k < 256 ? tw[k] : make_float2(-1.f, 0.f);
          const float xr = er + (orr * w.x - oi * w.y);
          const float xi = ei + (orr * w.y + oi * w.x);
          if (MODE == 0) p.spec[cf * kBins + k] = make_float2(xr, xi);
          if (MODE != 2) msum += sqrtf(xr * xr + xi * xi);
          if (MODE == 2 && k >= 1) {
            row[(k - 1) * 2 * p.nch + c] = __fdiv_rn(xr, den);
            row[(k - 1) * 2 * p.nch + p.nch + c] = __fdiv_rn(xi, den);
          }
        }
      }
      if (MODE != 2) {
        msum = wave_sum(msum);
        if (lane == 0) p.magsum[cf] = msum;
      }
      wave_lds_sync();                                         // z is reused by this wave's next channel
    }
    if (MODE == 2) {
      __syncthreads();
      float* o = p.x + (b * p.nt + t) * (long long)(kNF * 2 * p.nch);
      const int n = kNF * 2 * p.nch;                           // multiple of 4; the row starts on a 16-byte boundary
      for (int i = 4 * tid; i < n; i += 4 * kRowWaves * 64)
        *reinterpret_cast<float4*>(o + i) = *reinterpret_cast<const float4*>(row + i);
    }
  }
}

int rows_waves(int nch) { return nch <= 4 ? 4 : (nch <= 8 ? 8 : 16); }
// the frame-row kernel fetches 8 samples per thread: a frame's nch * 512 samples must fit 16 waves x 64 x 8 = 8192
constexpr int kRowsMaxCh = 16;

size_t rows_lds_bytes(int nch, int mode) {
  return (size_t)(768 + 512 + rows_waves(nch) * 512 + nch * kXsStride + (mode == 2 ? kNF * 2 * nch : 0)) * sizeof(float);
}

template <int MODE, int NW>
int launch_rows_nw(const RowParams& p, const char* name, hipStream_t st) {
  const size_t lds = rows_lds_bytes(p.nch, MODE);
  const long long nframes = (long long)p.nb * p.nt;
  // persistent: as many workgroups as fit the chip at once (LDS- and wave-limited), each walking its share of frames
  const int per_cu_lds = (int)((160 * 1024) / lds), per_cu_waves = 32 / NW;
  const int per_cu = per_cu_lds < per_cu_waves ? (per_cu_lds > 0 ? per_cu_lds : 1) : per_cu_waves;
  long long nblk = (long long)fnssl::device_cus() * per_cu;
  if (nblk > nframes) nblk = nframes;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stft_rows_kernel<MODE, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fnssl::TimedLaunch tl(name, st);
  hipLaunchKernelGGL((stft_rows_kernel<MODE, NW>), dim3((unsigned)nblk), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("stft_rows_kernel");
  return FNSSL_OK;
}

template <int MODE>
int launch_rows(const RowParams& p, const char* name, hipStream_t st) {
  FNSSL_REQUIRE(p.nch <= kRowsMaxCh && rows_lds_bytes(p.nch, MODE) <= 160 * 1024,
                "front end: the frame-row kernels take at most %d channels (%d given)", kRowsMaxCh, p.nch);
  FNSSL_REQUIRE((long long)p.nb * p.nt < (1ll << 31), "front end: too many frames");
  switch (rows_waves(p.nch)) {
    case 4: return launch_rows_nw<MODE, 4>(p, name, st);
    case 8: return launch_rows_nw<MODE, 8>(p, name, st);
    default: return launch_rows_nw<MODE, 16>(p, name, st);
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
  RowParams p{};
  p.sig = sig;
  p.nb = nb; p.nch = nch; p.nt = nt; p.ns = ns; p.hop = hop; p.center = center ? 1 : 0;
  p.sb = sb; p.sn = sn; p.sc = sc;
  p.spec = reinterpret_cast<float2*>(spec);
  p.magsum = magsum;
  if (magsum && nch <= kRowsMaxCh && rows_lds_bytes(nch, 0) <= 160 * 1024 && !fnssl::tune(FNSSL_TUNE_STFT_PER_FRAME))
    return launch_rows<0>(p, "stft", fnssl::as_stream(stream));
  // fallback (no magnitude sums wanted, or more channels than the frame image holds): one wave per (b, c, t)
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

int fnssl_array_frontend(const float* sig, int nb, int ns, int nch, long long sb, long long sn, long long sc, int hop,
                         int center, const float* coef_a, const float* coef_b, float eps, float* magsum, float* mu, float* x,
                         void* stream) {
  FNSSL_REQUIRE(nb > 0 && nch > 0, "array_frontend: empty batch (nb %d, nch %d)", nb, nch);
  FNSSL_REQUIRE(hop > 0 && hop <= kWin, "array_frontend: hop %d outside 1..%d", hop, kWin);
  const int nt = fnssl_num_frames_ex(ns, hop, center);
  FNSSL_REQUIRE(nt > 0, "array_frontend: signal of %d samples is too short", ns);
  FNSSL_REQUIRE(sig && coef_a && coef_b && magsum && mu && x, "array_frontend: null pointer");
  FNSSL_REQUIRE(nch <= kRowsMaxCh && rows_lds_bytes(nch, 2) <= 160 * 1024,
                "array_frontend: at most %d channels (%d given): use fnssl_stft_ex + fnssl_array_features", kRowsMaxCh, nch);
  hipStream_t st = fnssl::as_stream(stream);
  RowParams p{};
  p.sig = sig;
  p.nb = nb; p.nch = nch; p.nt = nt; p.ns = ns; p.hop = hop; p.center = center ? 1 : 0;
  p.sb = sb; p.sn = sn; p.sc = sc;
  p.magsum = magsum;
  p.mu = mu;
  p.eps = eps;
  p.x = x;
  {
    const int rc = launch_rows<1>(p, "stft", st);
    if (rc != FNSSL_OK) return rc;
  }
  {
    fnssl::TimedLaunch tl("ema", st);
    hipLaunchKernelGGL(ema_array_kernel, dim3((nb + kEmaWaves - 1) / kEmaWaves), dim3(kEmaWaves * 64), 0, st, magsum,
                       coef_a, coef_b, nb, nch, nt, mu);
    FNSSL_CHECK_LAUNCH("ema_array_kernel");
  }
  return launch_rows<2>(p, "pack", st);
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
