// IPDnet2 / OnlineSpatialNet on gfx950 (SURVEY.md 8 rows a13 + f4; reference IPDnet2/IPDnet2.py).
//
// The network is ~25 MFLOP per input frame spread over many SMALL dense layers (96 -> 96 grouped k5, 96 -> 8,
// 96 -> 384, 192 -> 40, 192 -> 96 ...) plus LayerNorms, poolings and one sequential scan — nothing like the
// LSTM recurrences next door.  fp32 MFMA is only 2x the packed-fp32 vector rate on CDNA4 and these layer
// shapes (12-wide groups, K = 60) would waste a quarter of every 16x16 tile, so the formulation here is
//
//     one THREAD = one time-frequency point; the point's 96 channels live in its registers;
//     weights are WAVE-UNIFORM: read through the scalar cache (s_load_dwordx8/16) and consumed as the
//     SGPR operand of v_pk_fma_f32 — no LDS or vector-memory traffic for weights at all;
//     LayerNorm is a reduction over the thread's own registers (no cross-lane step);
//     neighbours along frequency (the k = 5 grouped conv, the Linear over F) are exchanged through small
//     LDS tiles, one 12-channel group at a time, double-buffered (one barrier per group);
//     along time only the selective scan is sequential: one thread per (sequence, inner channel) keeps the
//     16 states in registers, B_t / C_t / dt_t arrive as wave-uniform scalars.
//
// Every kernel takes strided [B, T, F, 96] views (channels contiguous), so the reference's permutes
// (:222-233, :235-253, :333-335, :355-364) are never materialised and the residual adds, the frequency /
// time poolings and the activations are fused into the producing kernel.
//
// That first formulation survives as the scalar-operand kernels below (FNSSL_SN_SCALAR=1: the A/B reference the
// tests pin to the oracle and to the reference fixtures).  What runs by default is the second one, further down
// ("Dense layers on the matrix pipe"): a wave owns 16 points, the layer's weights sit in LDS as MFMA A operands
// for the whole launch, exact fp32 (v_mfma_f32_16x16x4_f32) or — FNSSL_PRECISION_BF16, BASELINE config 5 as
// written — bf16 operands with fp32 accumulation (v_mfma_f32_16x16x32_bf16); taps along time (encoder, the
// depthwise conv in front of x_proj) are DPP row shifts between the 16 lanes of a tile.  DESIGN.md section 10
// has the measurements that led from one to the other.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "tuning.h"

namespace {

constexpr int H = 96;      // dim_hidden
constexpr int HS = 8;      // dim_squeeze
constexpr int NG = 8;      // f-conv groups
constexpr int CG = 12;     // channels per group
constexpr int KF = 5;      // f-conv taps
constexpr int KE = 5;      // encoder taps
constexpr int E = 192;     // Mamba d_inner
constexpr int NST = 16;    // Mamba d_state
constexpr int RK = 6;      // Mamba dt_rank
constexpr int KC = 4;      // Mamba d_conv
constexpr int XP = 40;     // x_proj outputs (6 + 16 + 16) padded to a float4 multiple
constexpr int DO = 16;     // dim_output
constexpr float kEps = 1e-5f;

constexpr int kFconvTile = 4992;   // floats per buffer: max over nf in [8, 256] of (256/nf) * (nf + 4) * 13
constexpr int kFullTile = 2304;    // max over nf of (256/nf) * (nf * 8 + 4)

// x * sigmoid(x) with the reciprocal instruction (1 ulp) instead of the IEEE division sequence (10 operations): the
// unsqueeze of the full-band branch alone applies it 96 times per point
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ void load_row(float (&x)[H], const float* __restrict__ p) {
  const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int i = 0; i < H / 4; ++i) {
    const float4 v = p4[i];
    x[4 * i] = v.x;
    x[4 * i + 1] = v.y;
    x[4 * i + 2] = v.z;
    x[4 * i + 3] = v.w;
  }
}

__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&x)[H]) {
  float4* p4 = reinterpret_cast<float4*>(p);
#pragma unroll
  for (int i = 0; i < H / 4; ++i) p4[i] = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
}

// nn.LayerNorm statistics over the thread's own 96 channels (biased variance, eps inside the root).
__device__ __forceinline__ void ln_stats(const float (&x)[H], float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < H; ++c) s += x[c];
  mean = s * (1.f / H);
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < H; ++c) {
    const float d = x[c] - mean;
    v = fmaf(d, d, v);
  }
  rstd = 1.f / sqrtf(v * (1.f / H) + kEps);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm alone (arch/base/norm.py:11-27): one wave per row, wavefront reductions.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_layernorm_kernel(const float* __restrict__ x, long long rows, int h, const float* __restrict__ w,
                    const float* __restrict__ b, float eps, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * h;
  float s = 0.f;
  for (int c = lane; c < h; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)h;
  float v = 0.f;
  for (int c = lane; c < h; c += 64) {
    const float d = xr[c] - mean;
    v = fmaf(d, d, v);
  }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)h + eps);
  for (int c = lane; c < h; c += 64) y[row * h + c] = (xr[c] - mean) * rstd * w[c] + b[c];
}

// ---------------------------------------------------------------------------------------------------------
// Encoder: CausalConv1d(cin -> 96, k 5) along time (IPDnet2.py:66-76, :335).  Lanes run along t.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_encoder_kernel(const float* __restrict__ x, long long sb, long long sc, long long sf, long long st, int cin,
                  int nf, int nt, long long npts, const float* __restrict__ wT, const float* __restrict__ bias,
                  const float* __restrict__ state_in, float* __restrict__ out, long long o_sb, long long o_st,
                  long long o_sf) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts) return;
  const int t = (int)(p % nt);
  const int f = (int)((p / nt) % nf);
  const long long b = p / ((long long)nt * nf);
  float acc[H];
#pragma unroll
  for (int o = 0; o < H; ++o) acc[o] = bias[o];
  // the taps of channel c + 1 are requested before the 480 FMAs of channel c (the compiler does not move loads
  // across the back edge; without this every trip starts with a full memory round trip)
  auto taps = [&](int c, float (&v)[KE]) {
    const float* xc = x + b * sb + c * sc + f * sf;
    const float* sc_in = state_in ? state_in + ((b * cin + c) * nf + f) * (KE - 1) : nullptr;
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int tt = t + k - (KE - 1);
      v[k] = tt >= 0 ? xc[tt * st] : (sc_in ? sc_in[(KE - 1) + tt] : 0.f);
    }
  };
  float vn[KE];
  taps(0, vn);
  for (int c = 0; c < cin; ++c) {
    float v[KE];
#pragma unroll
    for (int k = 0; k < KE; ++k) v[k] = vn[k];
    if (c + 1 < cin) taps(c + 1, vn);
    const float* wc = wT + (long long)c * KE * H;
#pragma unroll
    for (int k = 0; k < KE; ++k)
#pragma unroll
      for (int o = 0; o < H; ++o) acc[o] = fmaf(wc[k * H + o], v[k], acc[o]);
  }
  store_row(out + b * o_sb + t * o_st + f * o_sf, acc);
}

// The carried state of the encoder: the last 4 input frames of every (b, c, f) row (in place is fine).
__global__ void __launch_bounds__(256)
sn_encoder_state_kernel(const float* __restrict__ x, long long sb, long long sc, long long sf, long long st,
                        int cin, int nf, int nt, long long nrows, const float* state_in, float* state_out) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;   // (b, c, f)
  if (r >= nrows) return;
  const int f = (int)(r % nf);
  const int c = (int)((r / nf) % cin);
  const long long b = r / ((long long)nf * cin);
  const float* xc = x + b * sb + c * sc + f * sf;
  float v[KE - 1];
#pragma unroll
  for (int j = 0; j < KE - 1; ++j) {
    const int src = nt - (KE - 1) + j;
    v[j] = src >= 0 ? xc[src * st] : (state_in ? state_in[r * (KE - 1) + (KE - 1) + src] : 0.f);
  }
#pragma unroll
  for (int j = 0; j < KE - 1; ++j) state_out[r * (KE - 1) + j] = v[j];
}

// ---------------------------------------------------------------------------------------------------------
// x (+)= PReLU(Conv1d_grouped_k5(LayerNorm(x))) along F, optional AvgPool over F (IPDnet2.py:146-153,222-233).
// One block = 256 points = 256/nf whole frames; thread = point.
// ---------------------------------------------------------------------------------------------------------
template <int POOL>
__global__ void __launch_bounds__(256)
sn_fconv_kernel(fnssl_btf_view xv, int nt, int nf, int lg_nf, long long nframes, fnssl_sn_fconv_w w, int residual,
                float* out, long long o_sb, long long o_st, long long o_sf) {
  __shared__ float tile[2][kFconvTile];
  const int tid = threadIdx.x;
  const int fr = tid >> lg_nf, f = tid & (nf - 1);
  const long long frame = (long long)blockIdx.x * (256 >> lg_nf) + fr;
  const bool valid = frame < nframes;
  const long long b = frame / nt;
  const int t = (int)(frame % nt);
  float x[H];
  if (valid) {
    load_row(x, xv.p + b * xv.sb + t * xv.st + f * xv.sf);
  } else {
#pragma unroll
    for (int c = 0; c < H; ++c) x[c] = 0.f;
  }
  const int rows = nf + 4;                     // 2 zero rows either side = the 'same' padding
  if (f < 2) {
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
      for (int ci = 0; ci < CG; ++ci) {
        tile[buf][(fr * rows + f) * 13 + ci] = 0.f;
        tile[buf][(fr * rows + nf + 2 + f) * 13 + ci] = 0.f;
      }
  }
  float mean, rstd;
  ln_stats(x, mean, rstd);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float* tb = tile[g & 1];
    float* mine = tb + (fr * rows + f + 2) * 13;
#pragma unroll
    for (int ci = 0; ci < CG; ++ci)
      mine[ci] = (x[g * CG + ci] - mean) * rstd * w.ln_w[g * CG + ci] + w.ln_b[g * CG + ci];
    __syncthreads();   // also orders this buffer's previous readers (iteration g - 2) before the next writers
    float acc[CG];
#pragma unroll
    for (int o = 0; o < CG; ++o) acc[o] = w.bias[g * CG + o];
    const float* wg = w.wT + g * (KF * CG * CG);
    const float* base = tb + (fr * rows + f) * 13;
    // rolled on purpose: a fully unrolled body lets the scheduler hoist all 720 scalar weight loads of the group
    // (5760 for the kernel) and spill SGPRs by the thousand; 48 weights per trip fit the scalar file
#pragma unroll 1
    for (int j = 0; j < KF * CG; j += 4) {
      const int tap = j / CG, ci = j - tap * CG;
      const float* bj = base + tap * 13 + ci;
      const float v0 = bj[0], v1 = bj[1], v2 = bj[2], v3 = bj[3];
      const float* wj = wg + j * CG;
#pragma unroll
      for (int o = 0; o < CG; ++o) acc[o] = fmaf(wj[o], v0, acc[o]);
#pragma unroll
      for (int o = 0; o < CG; ++o) acc[o] = fmaf(wj[CG + o], v1, acc[o]);
#pragma unroll
      for (int o = 0; o < CG; ++o) acc[o] = fmaf(wj[2 * CG + o], v2, acc[o]);
#pragma unroll
      for (int o = 0; o < CG; ++o) acc[o] = fmaf(wj[3 * CG + o], v3, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < CG; ++o) {
      float a = acc[o];
      a = a >= 0.f ? a : w.prelu[g * CG + o] * a;
      x[g * CG + o] = residual ? x[g * CG + o] + a : a;
    }
  }
  if (POOL > 1) {
#pragma unroll
    for (int c = 0; c < H; ++c) {
      float v = x[c];
      v += __shfl_xor(v, 1, 64);
      if (POOL == 8) {
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
      }
      x[c] = v * (1.f / POOL);
    }
  }
  if (valid && (f & (POOL - 1)) == 0) store_row(out + b * o_sb + t * o_st + (f / POOL) * o_sf, x);
}

// ---------------------------------------------------------------------------------------------------------
// x (+)= SiLU(unsqueeze(Linear_over_F(SiLU(squeeze(LayerNorm(x))))))   (IPDnet2.py:150,235-253)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_full_kernel(fnssl_btf_view xv, int nt, int nf, int lg_nf, long long nframes, fnssl_sn_full_w w, int residual,
               float* out, long long o_sb, long long o_st, long long o_sf) {
  __shared__ __attribute__((aligned(16))) float S[kFullTile];
  const int tid = threadIdx.x;
  const int fr = tid >> lg_nf, f = tid & (nf - 1);
  const long long frame = (long long)blockIdx.x * (256 >> lg_nf) + fr;
  const bool valid = frame < nframes;
  const long long b = frame / nt;
  const int t = (int)(frame % nt);
  float x[H];
  if (valid) {
    load_row(x, xv.p + b * xv.sb + t * xv.st + f * xv.sf);
  } else {
#pragma unroll
    for (int c = 0; c < H; ++c) x[c] = 0.f;
  }
  float mean, rstd;
  ln_stats(x, mean, rstd);
  float s[HS];
#pragma unroll
  for (int q = 0; q < HS; ++q) s[q] = w.bs[q];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float ln = (x[h] - mean) * rstd * w.ln_w[h] + w.ln_b[h];
#pragma unroll
    for (int q = 0; q < HS; ++q) s[q] = fmaf(w.wsT[h * HS + q], ln, s[q]);
  }
  const int fstride = nf * HS + 4;             // +4: frames of one wave land in different banks
  float4* mine = reinterpret_cast<float4*>(S + fr * fstride + f * HS);
  mine[0] = make_float4(silu_f(s[0]), silu_f(s[1]), silu_f(s[2]), silu_f(s[3]));
  mine[1] = make_float4(silu_f(s[4]), silu_f(s[5]), silu_f(s[6]), silu_f(s[7]));
  __syncthreads();
  float y[HS];
  const float bfv = w.bf[f];
#pragma unroll
  for (int q = 0; q < HS; ++q) y[q] = bfv;
  const float4* sf = reinterpret_cast<const float4*>(S + fr * fstride);
  for (int f2 = 0; f2 < nf; ++f2) {            // Linear over F: y[f] = sum_f2 W[f][f2] s[f2]
    const float wv = w.wfT[f2 * nf + f];
    const float4 a = sf[f2 * 2], c = sf[f2 * 2 + 1];
    y[0] = fmaf(wv, a.x, y[0]);
    y[1] = fmaf(wv, a.y, y[1]);
    y[2] = fmaf(wv, a.z, y[2]);
    y[3] = fmaf(wv, a.w, y[3]);
    y[4] = fmaf(wv, c.x, y[4]);
    y[5] = fmaf(wv, c.y, y[5]);
    y[6] = fmaf(wv, c.z, y[6]);
    y[7] = fmaf(wv, c.w, y[7]);
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float a = w.bu[h];
#pragma unroll
    for (int q = 0; q < HS; ++q) a = fmaf(w.wuT[q * H + h], y[q], a);
    a = silu_f(a);
    x[h] = residual ? x[h] + a : a;
  }
  if (valid) store_row(out + b * o_sb + t * o_st + f * o_sf, x);
}

// ---------------------------------------------------------------------------------------------------------
// Mamba, phase 1: xz[s, t, 0:384] = in_proj(LayerNorm(x))   (sequence s = b*nf + f; lanes run along t)
// blockIdx.y = output quarter (96 of the 384 outputs), so the weights stay wave-uniform.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_mamba_in_kernel(fnssl_btf_view xv, int nt, int nf, long long npts, const float* __restrict__ ln_w,
                   const float* __restrict__ ln_b, const float* __restrict__ winT, float* __restrict__ xz) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts) return;
  const int t = (int)(p % nt);
  const int f = (int)((p / nt) % nf);
  const long long b = p / ((long long)nt * nf);
  float x[H];
  load_row(x, xv.p + b * xv.sb + t * xv.st + f * xv.sf);
  float mean, rstd;
  ln_stats(x, mean, rstd);
#pragma unroll
  for (int h = 0; h < H; ++h) x[h] = (x[h] - mean) * rstd * ln_w[h] + ln_b[h];
  const int q = blockIdx.y;
  float4* dst = reinterpret_cast<float4*>(xz + p * (2 * E) + q * 96);
#pragma unroll 1
  for (int j = 0; j < 6; ++j) {
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    const float* wj = winT + q * 96 + j * 16;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int o = 0; o < 16; ++o) acc[o] = fmaf(wj[h * (2 * E) + o], x[h], acc[o]);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[j * 4 + i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Dense layers on the matrix pipe.  The scalar-operand formulation above is latency-bound where a launch has few
// points (layers 1..7 run on 51 k points = 800 waves: one wave walks 96 x 96 weights through scalar loads that return
// out of order, i.e. in batches behind s_waitcnt lgkmcnt(0) — 130 us for 8 us of FMAs), so the projections of the
// Mamba block and the encoder also exist as v_mfma_f32_16x16x4_f32 kernels (exact fp32, same flop rate as packed FMA):
//   a wave owns 16 points; lane (n, kq) holds the K/4 channels kq*K/4 .. of point n (the B operand of every k-step);
//   the weight matrix sits in LDS for the whole launch as A operands — tile j (16 outputs) x group g (4 k-steps):
//   lane (i, kq) reads W[kq*K/4 + 4g .. +3][16 j + i] with one ds_read_b128 — filled once per workgroup;
//   D[output 4*og + r][point n]: a lane stores 4 consecutive outputs of its point (16 bytes).
// ---------------------------------------------------------------------------------------------------------
// q = p / d, r = p % d (p >= 0, d > 0) in 32-bit arithmetic whenever p fits: a 64-bit division by a run-time value is
// ~150 VALU instructions, and the three per point of the index split were most of the instruction stream of the
// matrix-pipe kernels (counters, profiles/r02/m_: 1 200 VALU instructions per 16-point tile of the encoder).
__device__ __forceinline__ long long divmod(long long p, int d, int& r) {
  if (p < (1ll << 31)) {
    const unsigned pu = (unsigned)p, q = pu / (unsigned)d;
    r = (int)(pu - q * (unsigned)d);
    return (long long)q;
  }
  const long long q = p / d;
  r = (int)(p - q * d);
  return q;
}

typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf_t __attribute__((ext_vector_type(4)));

// FNSSL_PRECISION_BF16 (BF = true): the same kernels with both operands of the product rounded to bf16 (round to
// nearest even) as they enter the matrix pipe, fp32 accumulation, fp32 tensors — one v_mfma_f32_16x16x32_bf16 takes the
// place of eight v_mfma_f32_16x16x4_f32.  The lane's K/4 channels stay where they are: MFMA m consumes the lane's
// values 8 m .. 8 m + 7 (zero past K/4), and the weight image holds the matching 8 weights per (lane, m) as one
// 16-byte item, so an output tile costs ceil(K/32) LDS reads and MFMAs instead of K/16 reads and K/4 MFMAs.
__device__ __forceinline__ v8bf_t pack8_bf16(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                             float a7) {
  const v4f_t lo = {a0, a1, a2, a3}, hi = {a4, a5, a6, a7};
  const v4bf_t l = __builtin_convertvector(lo, v4bf_t), h = __builtin_convertvector(hi, v4bf_t);
  return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 1-KiB items (64 lanes x 16 bytes) of one 16-output tile in the LDS weight image
template <int K, bool BF>
__host__ __device__ constexpr int w_items() {
  return BF ? (K / 4 + 7) / 8 : K / 16;
}
template <int K, int N, bool BF>
__host__ __device__ constexpr int w_lds_floats() {
  return (N / 16) * w_items<K, BF>() * 256;
}

// sk / so: element strides of the input (k) and output (o) index in the source matrix.  BD = threads of the workgroup
// (compile-time: the trips are fully unrolled, so all of a thread's loads — 4 per 16-byte item, up to 12 items — are in
// flight together.  Rolled, a trip was one L2 round trip and the 147 KB image of in_proj took 21 us per workgroup: half
// of a small launch.)  LDS order, one conflict-free 16-byte write per item; an item's 4 values are 4 consecutive k of one
// output, and the 16 lanes of an output tile read 64 contiguous bytes of each source row (the matrix is L2-resident).
template <int K, int N, int BD, bool BF = false>
__device__ __forceinline__ void fill_w_lds_strided(float* lds, const float* __restrict__ w, int sk, int so, int kvalid,
                                                   int nvalid) {
  static_assert(K % 16 == 0 && N % 16 == 0, "whole k-step groups and output tiles");
  constexpr int KQ = K / 4, G = w_items<K, BF>(), TOTAL = (N / 16) * G * 64, TRIPS = (TOTAL + BD - 1) / BD;
  constexpr int PER = BF ? 8 : 4;                                    // k values of one item
  float v[TRIPS][PER];
#pragma unroll
  for (int u = 0; u < TRIPS; ++u) {
    const int idx = (int)threadIdx.x + u * BD;
    const int lane = idx & 63, g = (idx >> 6) % G, j = (idx >> 6) / G;
    const int kl = PER * g, k0 = (lane >> 4) * KQ + kl, o = 16 * j + (lane & 15);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      // unconditional load from a clamped address, then the choice: a conditional load is a branch and a wait per element
      const bool ok = idx < TOTAL && kl + i < KQ && k0 + i < kvalid && o < nvalid;
      const float x = w[ok ? (long long)(k0 + i) * sk + (long long)o * so : 0];
      v[u][i] = ok ? x : 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < TRIPS; ++u) {
    const int idx = (int)threadIdx.x + u * BD;
    if (idx < TOTAL) {
      if constexpr (BF)
        *reinterpret_cast<v8bf_t*>(lds + idx * 4) =
            pack8_bf16(v[u][0], v[u][1], v[u][2], v[u][3], v[u][PER - 4], v[u][PER - 3], v[u][PER - 2], v[u][PER - 1]);
      else
        *reinterpret_cast<float4*>(lds + idx * 4) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
}

template <int K, int N, int BD, bool BF = false>
__device__ __forceinline__ void fill_w_lds(float* lds, const float* __restrict__ wT, int ldw, int kvalid, int nvalid) {
  fill_w_lds_strided<K, N, BD, BF>(lds, wT, ldw, 1, kvalid, nvalid);
}

// lane n of a 16-lane row reads lane n - j (row_shr:j = 0x110 + j) or n + j (row_shl:j = 0x100 + j); 0 outside the row
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// bf16: the lane's K/4 values as ceil(K/32) packed operands (done once when several calls share them)
template <int K>
__device__ __forceinline__ void pack_operand(const float (&a)[K / 4], v8bf_t (&b)[w_items<K, true>()]) {
  constexpr int KQ = K / 4;
#pragma unroll
  for (int m = 0; m < w_items<K, true>(); ++m) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = 8 * m + e < KQ ? a[8 * m + e < KQ ? 8 * m + e : 0] : 0.f;
    b[m] = pack8_bf16(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
  }
}
template <int K, int NJ>
__device__ __forceinline__ void mfma_tiles_packed(const v8bf_t (&b)[w_items<K, true>()], const float* ldsw_lane, int j0,
                                                  v4f_t (&acc)[NJ]) {
  constexpr int G = w_items<K, true>();
#pragma unroll
  for (int m = 0; m < G; ++m)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const v8bf_t w = *reinterpret_cast<const v8bf_t*>(ldsw_lane + ((j0 + jj) * G + m) * 256);
      acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[m], acc[jj], 0, 0, 0);
    }
}

// acc[jj] += W-tile (j0 + jj) . x   for NJ output tiles at once (independent accumulators keep the pipe busy)
template <int K, int NJ, bool BF = false>
__device__ __forceinline__ void mfma_tiles(const float (&a)[K / 4], const float* ldsw_lane, int j0, v4f_t (&acc)[NJ]) {
  constexpr int G = w_items<K, BF>();
  if constexpr (BF) {
    constexpr int KQ = K / 4;
#pragma unroll
    for (int m = 0; m < G; ++m) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = 8 * m + e < KQ ? a[8 * m + e < KQ ? 8 * m + e : 0] : 0.f;
      const v8bf_t b = pack8_bf16(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const v8bf_t w = *reinterpret_cast<const v8bf_t*>(ldsw_lane + ((j0 + jj) * G + m) * 256);
        acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, acc[jj], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      v4f_t w[NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) w[jj] = *reinterpret_cast<const v4f_t*>(ldsw_lane + ((j0 + jj) * G + g) * 256);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[jj][i], a[4 * g + i], acc[jj], 0, 0, 0);
      // keep the scheduler from hoisting every group's LDS reads to the top (it spills the operand registers)
      if ((g & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Mamba, phase 1 on the matrix pipe: xz[p, 0:384] = in_proj(LayerNorm(x_p)),  p = (b*nf + f)*nt + t
template <bool BF>
__global__ void __launch_bounds__(1024)
sn_mamba_in_mfma_kernel(fnssl_btf_view xv, int nt, int nf, long long npts, const float* __restrict__ ln_w,
                        const float* __restrict__ ln_b, const float* __restrict__ winT, float* __restrict__ xz) {
  extern __shared__ __attribute__((aligned(16))) float ldsw[];     // 24 tiles x 6 groups x 1 KiB
  fill_w_lds<H, 2 * E, 1024, BF>(ldsw, winT, 2 * E, H, 2 * E);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const float* ldsw_lane = ldsw + lane * 4;
  const long long ntiles = (npts + 15) / 16;
  float lw[H / 4], lb[H / 4];
#pragma unroll
  for (int i = 0; i < H / 4; ++i) {
    lw[i] = ln_w[kq * (H / 4) + i];
    lb[i] = ln_b[kq * (H / 4) + i];
  }
  for (long long tile = (long long)blockIdx.x * 16 + w; tile < ntiles; tile += (long long)gridDim.x * 16) {
    const long long p = tile * 16 + n;
    const long long pc = p < npts ? p : npts - 1;
    int t, f;
    const long long b = divmod(divmod(pc, nt, t), nf, f);
    const float4* row = reinterpret_cast<const float4*>(xv.p + b * xv.sb + t * xv.st + f * xv.sf + kq * (H / 4));
    float a[H / 4];
#pragma unroll
    for (int i = 0; i < H / 16; ++i) {
      const float4 v = row[i];
      a[4 * i] = v.x;
      a[4 * i + 1] = v.y;
      a[4 * i + 2] = v.z;
      a[4 * i + 3] = v.w;
    }
    // LayerNorm over the point's 96 channels = the lane's 24 + the three lanes 16, 32, 48 further on
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < H / 4; ++i) sum += a[i];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / H);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < H / 4; ++i) {
      const float d = a[i] - mean;
      var = fmaf(d, d, var);
    }
    var += __shfl_xor(var, 16, 64);
    var += __shfl_xor(var, 32, 64);
    const float rstd = 1.f / sqrtf(var * (1.f / H) + kEps);
#pragma unroll
    for (int i = 0; i < H / 4; ++i) a[i] = (a[i] - mean) * rstd * lw[i] + lb[i];
    float* dst = xz + p * (2 * E) + 4 * kq;
#pragma unroll 1
    for (int j0 = 0; j0 < (2 * E) / 16; j0 += 4) {
      v4f_t acc[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[jj] = v4f_t{0.f, 0.f, 0.f, 0.f};
      mfma_tiles<H, 4, BF>(a, ldsw_lane, j0, acc);
      if (p < npts) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          *reinterpret_cast<float4*>(dst + 16 * (j0 + jj)) = make_float4(acc[jj][0], acc[jj][1], acc[jj][2], acc[jj][3]);
      }
    }
  }
}

// f-conv branch on the matrix pipe: x (+)= PReLU(Conv1d_grouped_k5(LayerNorm(x))) along F, optional AvgPool over F.
// A workgroup = 256 points = 256 / nf whole frames = 16 tiles of 16 consecutive bins.  Lane (n, q) of a tile owns bin n
// and, for q < 3, the channels 12 g + 4 q .. + 3 of every group g (32 registers): these are exactly the outputs
// D[4 q + r][n] of the group's product, so residual and pooling never leave the lane.  LayerNorm'd rows go to an LDS
// image [frame][bin + 2 (zero rows either side)][100] once; per group the lane gathers its 16 of the 64 (60 used)
// patch values (tap, channel) = the B operand, the group's [16 x 64] weights are A operands resident in LDS.
template <int POOL, bool BF>
__global__ void __launch_bounds__(1024)
sn_fconv_mfma_kernel(fnssl_btf_view xv, int nt, int nf, int lg_nf, long long nframes, long long nblk, fnssl_sn_fconv_w w,
                     int residual, float* out, long long o_sb, long long o_st, long long o_sf) {
  constexpr int YS = 100;                                          // row stride of the LDS image (floats)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int GW = w_lds_floats<64, 16, BF>();                   // a group's image: 4 (bf16: 2) items of 1 KiB
  float* ldw = lds;                                                // 8 groups
  float* par = lds + NG * GW;                                      // ln_w | ln_b | bias | prelu (L1 latency per use otherwise)
  float* ytile = par + 4 * H;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, q = lane >> 4;
  if constexpr (BF) {                                                // 8 groups x 2 items x 64 lanes = one item per thread
    const int gg = tid >> 7, idx = tid & 127;
    const int ln = idx & 63, m = idx >> 6;
    const int k0 = (ln >> 4) * 16 + 8 * m, o = ln & 15;
    const float* wg = w.wT + gg * (KF * CG * CG);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = k0 + i < KF * CG && o < CG;
      const float x = wg[ok ? (k0 + i) * CG + o : 0];
      v[i] = ok ? x : 0.f;
    }
    *reinterpret_cast<v8bf_t*>(ldw + gg * GW + idx * 4) = pack8_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int g = 0; g < NG; g += 4) {                                // 4 groups x 256 items = one item per thread
      const int gg = g + (tid >> 8);
      const int idx = tid & 255;
      const int ln = idx & 63, kg = idx >> 6;
      const int k0 = (ln >> 4) * 16 + 4 * kg, o = ln & 15;
      const float* wg = w.wT + gg * (KF * CG * CG);
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = k0 + i < KF * CG && o < CG;
        const float x = wg[ok ? (k0 + i) * CG + o : 0];
        v[i] = ok ? x : 0.f;
      }
      *reinterpret_cast<float4*>(ldw + gg * GW + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (tid < 4 * H) {
    const float* src = tid < H ? w.ln_w : (tid < 2 * H ? w.ln_b : (tid < 3 * H ? w.bias : w.prelu));
    par[tid] = src[tid % H];
  }
  __syncthreads();
  const int pt = 16 * wv + n;
  const int fr = pt >> lg_nf, f = pt & (nf - 1);
  const int rows = nf + 4;
  // patch element k = 16 q + i of a group = (tap k / 12, channel k % 12): 12 is a multiple of 4, so the lane's 16
  // elements are four aligned float4 of the image (offset tap * YS + channel from the bin's row - 2); k >= 60 is padding:
  // its weights are zero, so it reads the (initialised, finite) start of the row instead of being masked per value
  int poff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 16 * q + 4 * j;
    const int tap = (k * 171) >> 11;                               // k / 12 for k < 64
    poff[j] = k < KF * CG ? tap * YS + (k - CG * tap) : 0;
  }
  float* yrow = ytile + (fr * rows + f + 2) * YS + 4 * q;
  const float* ybase = ytile + (fr * rows + f) * YS;
  const float* ldw_lane = ldw + lane * 4;
  if (q < 3 && f < 2) {                                            // the 'same' padding: two zero rows either side, once
    float* z0 = ytile + (fr * rows + f) * YS + 4 * q;
    float* z1 = ytile + (fr * rows + nf + 2 + f) * YS + 4 * q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      *reinterpret_cast<float4*>(z0 + CG * g) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(z1 + CG * g) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto load_rows = [&](long long blk, float (&xr)[NG][4]) {
    const long long frame = blk * (256 >> lg_nf) + fr;
    const bool ok = blk < nblk && frame < nframes && q < 3;
    const long long fc = ok ? frame : 0;
    int fct;
    const long long fcb = divmod(fc, nt, fct);
    const float* row = xv.p + fcb * xv.sb + fct * xv.st + f * xv.sf + 4 * q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(row + CG * g);
      xr[g][0] = v.x;
      xr[g][1] = v.y;
      xr[g][2] = v.z;
      xr[g][3] = v.w;
    }
  };
  // persistent workgroup: the weight image is filled once; the next block's rows are requested before this block's
  // products, so a block costs its LDS passes and MFMAs, not an HBM round trip
  float xn[NG][4];
  load_rows(blockIdx.x, xn);
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    float xr[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) xr[g][r] = xn[g][r];
    const long long frame = blk * (256 >> lg_nf) + fr;
    const bool valid = frame < nframes;
    int t;
    const long long b = divmod(frame, nt, t);
    // LayerNorm over the bin's 96 channels: 32 in each of the lanes q = 0..2 (q = 3 holds zeros)
    // (two channels per instruction: the kernel is bound by instruction issue, counters in profiles/r02/m_)
    v2f_t sum2 = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) sum2 += v2f_t{xr[g][0], xr[g][1]} + v2f_t{xr[g][2], xr[g][3]};
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / H);
    const v2f_t mean2 = {mean, mean};
    v2f_t var2 = {0.f, 0.f};
    if (q < 3) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const v2f_t d = v2f_t{xr[g][2 * h], xr[g][2 * h + 1]} - mean2;
          var2 = __builtin_elementwise_fma(d, d, var2);
        }
    }
    float var = var2[0] + var2[1];
    var += __shfl_xor(var, 16, 64);
    var += __shfl_xor(var, 32, 64);
    const float rstd = 1.f / sqrtf(var * (1.f / H) + kEps);
    const v2f_t rstd2 = {rstd, rstd};
    if (q < 3) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float4 lw = *reinterpret_cast<const float4*>(par + CG * g + 4 * q);
        const float4 lb = *reinterpret_cast<const float4*>(par + H + CG * g + 4 * q);
        const v2f_t y0 = __builtin_elementwise_fma((v2f_t{xr[g][0], xr[g][1]} - mean2) * rstd2, v2f_t{lw.x, lw.y}, v2f_t{lb.x, lb.y});
        const v2f_t y1 = __builtin_elementwise_fma((v2f_t{xr[g][2], xr[g][3]} - mean2) * rstd2, v2f_t{lw.z, lw.w}, v2f_t{lb.z, lb.w});
        *reinterpret_cast<float4*>(yrow + CG * g) = make_float4(y0[0], y0[1], y1[0], y1[1]);
      }
    }
    __syncthreads();
    load_rows(blk + gridDim.x, xn);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float a[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(ybase + CG * g + poff[j]);
        a[4 * j] = v.x;
        a[4 * j + 1] = v.y;
        a[4 * j + 2] = v.z;
        a[4 * j + 3] = v.w;
      }
      v4f_t acc[1] = {v4f_t{0.f, 0.f, 0.f, 0.f}};
      mfma_tiles<64, 1, BF>(a, ldw_lane + g * GW, 0, acc);
      if (q < 3) {
        const float4 bv = *reinterpret_cast<const float4*>(par + 2 * H + CG * g + 4 * q);
        const float4 pv = *reinterpret_cast<const float4*>(par + 3 * H + CG * g + 4 * q);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const v2f_t v = v2f_t{acc[0][2 * h], acc[0][2 * h + 1]} + v2f_t{bb[2 * h], bb[2 * h + 1]};
          const v2f_t nv = v2f_t{pp[2 * h], pp[2 * h + 1]} * v;
          v2f_t o = {v[0] >= 0.f ? v[0] : nv[0], v[1] >= 0.f ? v[1] : nv[1]};
          if (residual) o += v2f_t{xr[g][2 * h], xr[g][2 * h + 1]};
          xr[g][2 * h] = o[0];
          xr[g][2 * h + 1] = o[1];
        }
      }
    }
    if (POOL > 1) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = xr[g][r];
          v += __shfl_xor(v, 1, 64);
          if (POOL == 8) {
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
          }
          xr[g][r] = v * (1.f / POOL);
        }
    }
    if (valid && q < 3 && (f & (POOL - 1)) == 0) {
      float* dst = out + b * o_sb + t * o_st + (f / POOL) * o_sf + 4 * q;
#pragma unroll
      for (int g = 0; g < NG; ++g)
        *reinterpret_cast<float4*>(dst + CG * g) = make_float4(xr[g][0], xr[g][1], xr[g][2], xr[g][3]);
    }
    __syncthreads();                                               // everybody is done reading the image
  }
}

// Full-band branch on the matrix pipe: x (+)= SiLU(unsqueeze(Linear_over_F(SiLU(squeeze(LayerNorm(x)))))).
// Same workgroup shape and channel ownership as the f-conv kernel (256 points = whole frames; lane (n, q) owns bin n and
// the channels 12 g + 4 q .. + 3, q < 3), three products:
//   squeeze   s[8]  = Ws . ln        K = 96 (the lane's 32 registers ARE its B operand; lane q = 3 holds zeros, K = 128),
//                                    N = 16 (8 used); D gives lane (bin, q < 2) the outputs 4 q .. + 3
//   over F    y[f'][(frame, c)] = sum_f Wf[f][f'] s[f][(frame, c)]   K = M = NF, 16 columns = 2 frames x 8 squeezed
//                                    channels per tile: always 8 (M-tile, column-tile) pairs, one per wave 0..7;
//                                    s goes through LDS transposed per frame ([frame][c][f], f contiguous = the B operand)
//   unsqueeze out[96] = Wu . y       K = 8 (16), 8 output tiles whose rows are permuted to the ownership order, so the
//                                    D fragment lands on the lane that holds the residual row
template <int NF, bool BF>
__global__ void __launch_bounds__(1024)
sn_full_mfma_kernel(fnssl_btf_view xv, int nt, long long nframes, long long nblk, fnssl_sn_full_w w, int residual,
                    float* out, long long o_sb, long long o_st, long long o_sf) {
  constexpr int LG = NF == 128 ? 7 : (NF == 64 ? 6 : (NF == 32 ? 5 : 4));
  constexpr int FPB = 256 / NF;                                     // frames per block
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldwf = lds;                                                // (NF/16)^2 KiB (bf16: NF/16 * ceil(NF/32))
  float* ldws = ldwf + w_lds_floats<NF, NF, BF>();                  // 8 KiB (bf16: 4)
  float* ldwu = ldws + w_lds_floats<128, 16, BF>();                 // 8 KiB
  float* par = ldwu + 8 * 256;                                      // ln_w 96 | ln_b 96 | bu 96 | bs 16 | bf NF
  float* st_img = par + 3 * H + 16 + NF;                            // [frame][8][NF]
  float* y_img = st_img + 256 * 8;                                  // [point][8]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, q = lane >> 4;
  fill_w_lds<NF, NF, 1024, BF>(ldwf, w.wfT, NF, NF, NF);
  if (tid < 512) {                                                  // squeeze image: 8 k-groups x 64 lanes
    const int ln = tid & 63, gk = tid >> 6, o = ln & 15, kq = ln >> 4;
    if constexpr (BF) {                                             // bf16: 4 items of 8 = the k-groups 2 m, 2 m + 1
      if (tid < 256) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool ok = kq < 3 && o < HS;
          const float x = w.wsT[ok ? (CG * (2 * gk + (i >> 2)) + 4 * kq + (i & 3)) * HS + o : 0];
          v[i] = ok ? x : 0.f;
        }
        *reinterpret_cast<v8bf_t*>(ldws + tid * 4) = pack8_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      }
    } else {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = kq < 3 && o < HS;
        const float x = w.wsT[ok ? (CG * gk + 4 * kq + i) * HS + o : 0];
        v[i] = ok ? x : 0.f;
      }
      *reinterpret_cast<float4*>(ldws + tid * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {                                                          // unsqueeze image: 8 output tiles x 64 lanes
    const int t2 = tid - 512, ln = t2 & 63, g = t2 >> 6, orow = ln & 15, kq = ln >> 4, og = orow >> 2, r = orow & 3;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = og < 3 && 4 * kq + i < HS;
      const float x = w.wuT[ok ? (4 * kq + i) * H + CG * g + 4 * og + r : 0];
      v[i] = ok ? x : 0.f;
    }
    if constexpr (BF)                                               // K = 16 is half of one bf16 k-step: 4 zeros follow
      *reinterpret_cast<v8bf_t*>(ldwu + t2 * 4) = pack8_bf16(v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f);
    else
      *reinterpret_cast<float4*>(ldwu + t2 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int i = tid; i < 3 * H + 16 + NF; i += 1024) {
    float v;
    if (i < H) v = w.ln_w[i];
    else if (i < 2 * H) v = w.ln_b[i - H];
    else if (i < 3 * H) v = w.bu[i - 2 * H];
    else if (i < 3 * H + 16) v = i - 3 * H < HS ? w.bs[i - 3 * H] : 0.f;
    else v = w.bf[i - 3 * H - 16];
    par[i] = v;
  }
  __syncthreads();
  const int pt = 16 * wv + n;
  const int fr = pt >> LG, f = pt & (NF - 1);
  const float* ldws_lane = ldws + lane * 4;
  const float* ldwu_lane = ldwu + lane * 4;
  const float* ldwf_lane = ldwf + lane * 4;
  auto load_rows = [&](long long blk, float (&xr)[NG][4]) {
    const long long frame = blk * FPB + fr;
    const bool ok = blk < nblk && frame < nframes && q < 3;
    const long long fc = ok ? frame : 0;
    int fct;
    const long long fcb = divmod(fc, nt, fct);
    const float* row = xv.p + fcb * xv.sb + fct * xv.st + f * xv.sf + 4 * q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(row + CG * g);
      xr[g][0] = v.x;
      xr[g][1] = v.y;
      xr[g][2] = v.z;
      xr[g][3] = v.w;
    }
  };
  float xn[NG][4];
  load_rows(blockIdx.x, xn);
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    float xr[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) xr[g][r] = xn[g][r];
    const long long frame = blk * FPB + fr;
    const bool valid = frame < nframes;
    int t;
    const long long b = divmod(frame, nt, t);
    // LayerNorm (32 channels in each of the lanes q = 0..2)
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) sum += (xr[g][0] + xr[g][1]) + (xr[g][2] + xr[g][3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / H);
    float var = 0.f;
    if (q < 3) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = xr[g][r] - mean;
          var = fmaf(d, d, var);
        }
    }
    var += __shfl_xor(var, 16, 64);
    var += __shfl_xor(var, 32, 64);
    const float rstd = 1.f / sqrtf(var * (1.f / H) + kEps);
    auto ln_row = [&](int g, float (&o)[4]) {
      const float4 lw = *reinterpret_cast<const float4*>(par + CG * g + 4 * (q < 3 ? q : 0));
      const float4 lb = *reinterpret_cast<const float4*>(par + H + CG * g + 4 * (q < 3 ? q : 0));
      const float m = q < 3 ? 1.f : 0.f;
      o[0] = ((xr[g][0] - mean) * rstd * lw.x + lb.x) * m;
      o[1] = ((xr[g][1] - mean) * rstd * lw.y + lb.y) * m;
      o[2] = ((xr[g][2] - mean) * rstd * lw.z + lb.z) * m;
      o[3] = ((xr[g][3] - mean) * rstd * lw.w + lb.w) * m;
    };
    // ---- squeeze -> SiLU -> transposed image [frame][c][f]
    {
      v4f_t acc[1] = {v4f_t{0.f, 0.f, 0.f, 0.f}};
      if constexpr (BF) {                                           // a pair of groups is packed and consumed at once:
#pragma unroll
        for (int m = 0; m < NG / 2; ++m) {                          // the 32 normalised values are never all live
          float lo[4], hi[4];
          ln_row(2 * m, lo);
          ln_row(2 * m + 1, hi);
          const v8bf_t bq = pack8_bf16(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
          const v8bf_t wq = *reinterpret_cast<const v8bf_t*>(ldws_lane + m * 256);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, bq, acc[0], 0, 0, 0);
        }
      } else {
        float a[32];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          float o[4];
          ln_row(g, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) a[4 * g + r] = o[r];
        }
        mfma_tiles<128, 1, false>(a, ldws_lane, 0, acc);
      }
      if (q < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          st_img[(fr * HS + 4 * q + r) * NF + f] = silu_f(acc[0][r] + par[3 * H + 4 * q + r]);
      }
    }
    __syncthreads();
    load_rows(blk + gridDim.x, xn);
    // ---- Linear over F: pair (M-tile mt, column tile ct) on wave 0..7; column n = (frame 2 ct + (n >> 3), c = n & 7)
    if (wv < 8) {
      constexpr int CT = 128 / NF;                                  // column tiles
      const int mt = wv / CT, ct = wv % CT;
      const int cfr = 2 * ct + (n >> 3), cc = n & 7;
      const float4* src = reinterpret_cast<const float4*>(st_img + (cfr * HS + cc) * NF + q * (NF / 4));
      v4f_t acc[1] = {v4f_t{0.f, 0.f, 0.f, 0.f}};
      if constexpr (BF) {
        constexpr int M = w_items<NF, true>();
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float4 lo = src[NF >= 32 ? 2 * m : 0];
          const float4 hi = NF >= 32 ? src[NF >= 32 ? 2 * m + 1 : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
          const v8bf_t bq = pack8_bf16(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
          const v8bf_t wq = *reinterpret_cast<const v8bf_t*>(ldwf_lane + (mt * M + m) * 256);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, bq, acc[0], 0, 0, 0);
        }
      } else {
        float bv[NF / 4];
#pragma unroll
        for (int i = 0; i < NF / 16; ++i) {
          const float4 v = src[i];
          bv[4 * i] = v.x;
          bv[4 * i + 1] = v.y;
          bv[4 * i + 2] = v.z;
          bv[4 * i + 3] = v.w;
        }
        mfma_tiles<NF, 1, false>(bv, ldwf_lane, mt, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fo = 16 * mt + 4 * q + r;
        y_img[(cfr * NF + fo) * HS + cc] = acc[0][r] + par[3 * H + 16 + fo];
      }
    }
    __syncthreads();
    // ---- unsqueeze -> SiLU -> residual
    {
      float yv[4];
      const float4 v = *reinterpret_cast<const float4*>(y_img + pt * HS + 4 * (q < 2 ? q : 0));
      const float m = q < 2 ? 1.f : 0.f;
      yv[0] = v.x * m;
      yv[1] = v.y * m;
      yv[2] = v.z * m;
      yv[3] = v.w * m;
#pragma unroll
      for (int g0 = 0; g0 < NG; g0 += 4) {
        v4f_t acc[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = v4f_t{0.f, 0.f, 0.f, 0.f};
        mfma_tiles<16, 4, BF>(yv, ldwu_lane, g0, acc);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float4 bu4 = *reinterpret_cast<const float4*>(par + 2 * H + CG * (g0 + jj) + 4 * (q < 3 ? q : 0));
          const float bb[4] = {bu4.x, bu4.y, bu4.z, bu4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float u = silu_f(acc[jj][r] + bb[r]);
            xr[g0 + jj][r] = residual ? xr[g0 + jj][r] + u : u;
          }
        }
      }
    }
    if (valid && q < 3) {
      float* dst = out + b * o_sb + t * o_st + f * o_sf + 4 * q;
#pragma unroll
      for (int g = 0; g < NG; ++g)
        *reinterpret_cast<float4*>(dst + CG * g) = make_float4(xr[g][0], xr[g][1], xr[g][2], xr[g][3]);
    }
    // the next block's images are written after its own barriers; st_img is rewritten only after the __syncthreads()
    // that follows this block's last read of it (above), y_img after the next block's first barrier
  }
}

// Mamba, phase 2a: u[p, e] = SiLU(conv4(xi))(p, e) over the taps t-3..t of xz (thread = point x channel quad).  u is
// what x_proj and the scan both consume; it lives in the scan's output buffer (the scan reads u_t before it writes y_t).
__global__ void __launch_bounds__(256)
sn_mamba_conv_kernel(const float* __restrict__ xz, int nt, long long npts, const float* __restrict__ conv_w,
                     const float* __restrict__ conv_b, const float* __restrict__ conv_state, float* __restrict__ u) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npts * (E / 4)) return;
  int e4, t;
  const long long p = divmod(idx, E / 4, e4);
  const int e0 = e4 * 4;
  const long long sq = divmod(p, nt, t);
  float4 v[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int tt = t - (KC - 1) + k;
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tt >= 0) {
      v[k] = *reinterpret_cast<const float4*>(xz + (sq * nt + tt) * (2 * E) + e0);
    } else if (conv_state) {
      v[k] = *reinterpret_cast<const float4*>(conv_state + (sq * (KC - 1) + (KC - 1) + tt) * E + e0);
    }
  }
  const float xi[KC][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                           {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
  const float4 cb = *reinterpret_cast<const float4*>(conv_b + e0);
  const float cbv[4] = {cb.x, cb.y, cb.z, cb.w};
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 cw = *reinterpret_cast<const float4*>(conv_w + (e0 + i) * KC);
    float a = cbv[i];
    a = fmaf(cw.x, xi[0][i], a);
    a = fmaf(cw.y, xi[1][i], a);
    a = fmaf(cw.z, xi[2][i], a);
    a = fmaf(cw.w, xi[3][i], a);
    r[i] = silu_f(a);
  }
  *reinterpret_cast<float4*>(u + p * E + e0) = make_float4(r[0], r[1], r[2], r[3]);
}

// Mamba, phase 2b on the matrix pipe: dbl[p, 0:40] = x_proj(u_p)
template <bool BF>
__global__ void __launch_bounds__(512)
sn_mamba_xproj_mfma_kernel(const float* __restrict__ u, long long npts, const float* __restrict__ wxT,
                           float* __restrict__ dbl) {
  constexpr int XN = 48;                                           // 40 outputs in 3 tiles
  extern __shared__ __attribute__((aligned(16))) float ldsw[];     // 3 tiles x 12 groups x 1 KiB
  fill_w_lds<E, XN, 512, BF>(ldsw, wxT, XP, E, XP);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const float* ldsw_lane = ldsw + lane * 4;
  const long long ntiles = (npts + 15) / 16;
  for (long long tile = (long long)blockIdx.x * 8 + w; tile < ntiles; tile += (long long)gridDim.x * 8) {
    const long long p = tile * 16 + n;
    const long long pc = p < npts ? p : npts - 1;
    const float4* ub = reinterpret_cast<const float4*>(u + pc * E + kq * (E / 4));
    float a[E / 4];
#pragma unroll
    for (int e4 = 0; e4 < E / 16; ++e4) {
      const float4 v = ub[e4];
      a[4 * e4] = v.x;
      a[4 * e4 + 1] = v.y;
      a[4 * e4 + 2] = v.z;
      a[4 * e4 + 3] = v.w;
    }
    v4f_t acc[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) acc[jj] = v4f_t{0.f, 0.f, 0.f, 0.f};
    mfma_tiles<E, 3, BF>(a, ldsw_lane, 0, acc);
    if (p < npts) {
      float* dst = dbl + p * XP + 4 * kq;
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
        if (16 * jj + 4 * kq < XP)
          *reinterpret_cast<float4*>(dst + 16 * jj) = make_float4(acc[jj][0], acc[jj][1], acc[jj][2], acc[jj][3]);
    }
  }
}

// Mamba, phases 2a + 2b in one pass (whole sequences, no carried taps): dbl[p, 0:40] = x_proj(SiLU(conv4(xi)))(p).
// A tile = 16 consecutive frames t0 .. t0 + 15 of ONE sequence with t0 = 13 k - 3: the lanes n >= 3 are the tile's 13
// outputs, the lanes 0..2 only carry the three frames before them — every tap of the depthwise conv is then the own
// frame of a left neighbour (DPP row shifts), frames before the start of the sequence load as zero, and nothing is
// masked.  u is NOT written: the scan recomputes it from xz (4 FMAs + SiLU per step, have_u = 0), which is cheaper
// than a 2 x 197-MB round trip and a launch.  Conv taps / bias of the lane's 48 channels come from an LDS image
// (four addresses per read: the 16 lanes of a row share kq).
template <bool BF>
__global__ void __launch_bounds__(512)
sn_mamba_convx_mfma_kernel(const float* __restrict__ xz, int nt, long long nseq, const float* __restrict__ conv_w,
                           const float* __restrict__ conv_b, const float* __restrict__ wxT, float* __restrict__ dbl) {
  constexpr int XN = 48, OUTS = 16 - (KC - 1);
  extern __shared__ __attribute__((aligned(16))) float ldsw[];
  float* ldcw = ldsw + w_lds_floats<E, XN, BF>();                   // [E][4] taps, then [E] bias
  float* ldcb = ldcw + E * KC;
  fill_w_lds<E, XN, 512, BF>(ldsw, wxT, XP, E, XP);
  for (int i = threadIdx.x; i < E * KC; i += 512) ldcw[i] = conv_w[i];
  if (threadIdx.x < E) ldcb[threadIdx.x] = conv_b[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const float* ldsw_lane = ldsw + lane * 4;
  const int tps = (nt + OUTS - 1) / OUTS;                           // tiles per sequence
  const long long ntiles = nseq * tps;
  for (long long tile = (long long)blockIdx.x * 8 + w; tile < ntiles; tile += (long long)gridDim.x * 8) {
    int k;
    const long long sq = divmod(tile, tps, k);                      // wave-uniform
    const int t = OUTS * k - (KC - 1) + n;
    const bool in = t >= 0 && t < nt;
    const long long p = sq * nt + (in ? t : 0);
    const float4* row = reinterpret_cast<const float4*>(xz + p * (2 * E) + kq * (E / 4));
    float4 xi[E / 16];
#pragma unroll
    for (int i = 0; i < E / 16; ++i) xi[i] = row[i];                // raw (a frame outside the sequence is zeroed below)
    float a[E / 4];
#pragma unroll
    for (int i = 0; i < E / 16; ++i) {
      const float x3[4] = {in ? xi[i].x : 0.f, in ? xi[i].y : 0.f, in ? xi[i].z : 0.f, in ? xi[i].w : 0.f};
      const float4 cb4 = *reinterpret_cast<const float4*>(ldcb + kq * (E / 4) + 4 * i);
      const float cbv[4] = {cb4.x, cb4.y, cb4.z, cb4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 cw = *reinterpret_cast<const float4*>(ldcw + (kq * (E / 4) + 4 * i + c) * KC);
        const float x2 = dpp_f<0x111>(x3[c]), x1 = dpp_f<0x112>(x3[c]), x0 = dpp_f<0x113>(x3[c]);   // frames t-1, t-2, t-3
        float u = cbv[c];
        u = fmaf(cw.x, x0, u);
        u = fmaf(cw.y, x1, u);
        u = fmaf(cw.z, x2, u);
        u = fmaf(cw.w, x3[c], u);
        a[4 * i + c] = silu_f(u);
      }
      asm volatile("" ::: "memory");                                // the 48 tap reads are not all hoisted to the top
    }
    v4f_t acc[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) acc[jj] = v4f_t{0.f, 0.f, 0.f, 0.f};
    mfma_tiles<E, 3, BF>(a, ldsw_lane, 0, acc);
    if (in && n >= KC - 1) {
      float* dst = dbl + p * XP + 4 * kq;
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
        if (16 * jj + 4 * kq < XP)
          *reinterpret_cast<float4*>(dst + 16 * jj) = make_float4(acc[jj][0], acc[jj][1], acc[jj][2], acc[jj][3]);
    }
  }
}

// Mamba, phase 4 on the matrix pipe: out = pool_T(x) + out_proj(pool_T(y)),  p = s*nt2 + t2.
// TP = the time pooling when it is 1 or 5 (0: any, taken from tp_rt): a run-time trip count left the pooled frames as
// 60 dependent load - wait - add round trips per tile (237 us for layer 0's pooled block against 141 us for the
// unpooled one with five times the tiles).
template <bool BF, int TP>
__global__ void __launch_bounds__(512)
sn_mamba_out_mfma_kernel(const float* __restrict__ ybuf, fnssl_btf_view xv, int nt, int nt2, int nf, int tp_rt,
                         long long nout, const float* __restrict__ woT, int residual, float* out, long long o_sb,
                         long long o_st, long long o_sf) {
  const int tp = TP ? TP : tp_rt;
  extern __shared__ __attribute__((aligned(16))) float ldsw[];     // 6 tiles x 12 groups x 1 KiB
  fill_w_lds<E, H, 512, BF>(ldsw, woT, H, E, H);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const float* ldsw_lane = ldsw + lane * 4;
  const long long ntiles = (nout + 15) / 16;
  const float inv = 1.f / (float)tp;
  for (long long tile = (long long)blockIdx.x * 8 + w; tile < ntiles; tile += (long long)gridDim.x * 8) {
    const long long p = tile * 16 + n;
    const long long pc = p < nout ? p : nout - 1;
    int t2, f;
    const long long sq = divmod(pc, nt2, t2);
    const long long b = divmod(sq, nf, f);
    const float* yb = ybuf + (sq * nt + (long long)t2 * tp) * E + kq * (E / 4);
    float a[E / 4];
#pragma unroll
    for (int e4 = 0; e4 < E / 16; ++e4) {
      float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < tp; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(yb + (long long)i * E + e4 * 4);
        yv.x += v.x;
        yv.y += v.y;
        yv.z += v.z;
        yv.w += v.w;
      }
      a[4 * e4] = yv.x * inv;
      a[4 * e4 + 1] = yv.y * inv;
      a[4 * e4 + 2] = yv.z * inv;
      a[4 * e4 + 3] = yv.w * inv;
      // loads in flight: 4 float4 per lane, pooled (tp of them per e4) 2 * tp — held there by a compiler-level memory
      // barrier (a scheduling barrier alone let all 60 be issued ahead of the first add: 240 registers)
      if (TP >= 4) {
        if (e4 % 2 == 1) asm volatile("" ::: "memory");
      } else if ((e4 & 3) == 3) {
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const float* xb = xv.p + b * xv.sb + (long long)t2 * tp * xv.st + f * xv.sf + 4 * kq;
    float* dst = out + b * o_sb + t2 * o_st + f * o_sf + 4 * kq;
    if (TP >= 4) asm volatile("" ::: "memory");
#pragma unroll
    for (int j0 = 0; j0 < H / 16; j0 += 3) {
      v4f_t acc[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) acc[jj] = v4f_t{0.f, 0.f, 0.f, 0.f};
      mfma_tiles<E, 3, BF>(a, ldsw_lane, j0, acc);
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        float4 o4 = make_float4(acc[jj][0], acc[jj][1], acc[jj][2], acc[jj][3]);
        if (residual) {
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int i = 0; i < tp; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(xb + (long long)i * xv.st + 16 * (j0 + jj));
            r.x += v.x;
            r.y += v.y;
            r.z += v.z;
            r.w += v.w;
          }
          o4 = make_float4(fmaf(r.x, inv, o4.x), fmaf(r.y, inv, o4.y), fmaf(r.z, inv, o4.z), fmaf(r.w, inv, o4.w));
        }
        if (p < nout) *reinterpret_cast<float4*>(dst + 16 * (j0 + jj)) = o4;
      }
      if (TP >= 4) asm volatile("" ::: "memory");                  // not both halves' 30 residual loads at once
    }
  }
}

// Encoder on the matrix pipe: K = cin * 5 taps (c-major) padded to KP (80 or 160: KP / 4 a multiple of the 5 taps);
// p = (b*nf + f)*nt + t
// PIPE: the whole-signal formulation (no carried frames, nt >= 4) — its own instantiation so that each path gets its own
// register allocation
template <int KP, bool BF, bool PIPE>
__global__ void __launch_bounds__(512, PIPE ? 4 : 1)                 // PIPE: two workgroups per CU = 4 waves per SIMD
sn_encoder_mfma_kernel(const float* __restrict__ x, long long sb, long long sc, long long sf, long long st, int cin,
                       int nf, int nt, long long npts, const float* __restrict__ wT, const float* __restrict__ bias,
                       const float* __restrict__ state_in, float* __restrict__ out, long long o_sb, long long o_st,
                       long long o_sf) {
  extern __shared__ __attribute__((aligned(16))) float ldsw[];     // 6 tiles x KP/16 groups x 1 KiB (+ PIPE: bias)
  fill_w_lds<KP, H, 512, BF>(ldsw, wT, H, cin * KE, H);
  float* ldsb = ldsw + w_lds_floats<KP, H, BF>();
  if (PIPE && threadIdx.x < H) ldsb[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const float* ldsw_lane = ldsw + lane * 4;
  const long long ntiles = (npts + 15) / 16;
  if constexpr (PIPE) {
    // Whole-signal case.  ONE load per channel (the lane's current frame) plus, in the lanes n < 4, the frame of the
    // point p - 4 (the halo of the tile); the other four taps of a lane are the current frames of its left neighbours,
    // which are its own earlier frames whenever t >= shift (points are consecutive along t) — DPP row shifts within
    // the tile's 16 lanes: 2 loads + 12 moves per channel instead of 5 loads.  The NEXT tile's loads are issued before
    // this tile's products and stores (raw values only: arithmetic on a loaded value would make the wave wait where
    // it is loaded); the bias comes from LDS and element offsets are 32-bit from the uniform base (the host checks the
    // tensor's extent), so nothing else in the loop uses the memory counter or needs sixteen 64-bit addresses.
    // (Tiles of one (b, f) row with the workgroup's waves on adjacent bins — 3-KB output runs instead of 98-KB-strided
    // 384-byte chunks — measured slower: 0.72 against 0.63 ms.)
    constexpr int NC = KP / 20;
    const int isc = (int)sc, ist = (int)st, lane_c = NC * kq;
    float cur[NC], halo[NC];
    int t = 0;
    float* dst = out;
    bool pok = false;
    auto fetch = [&](long long tile) {
      const long long p = tile * 16 + n;
      pok = p < npts;
      const long long pc = pok ? p : npts - 1;
      int f;
      const long long row = divmod(pc, nt, t);                      // (b, f) row of the point
      const long long b = divmod(row, nf, f);
      const int rowoff = (int)b * (int)sb + f * (int)sf + lane_c * isc;
      int off = rowoff + t * ist;
      int hoff = off - (KE - 1) * ist;                              // halo point p - 4: same row, or the end of the row before
      bool hok = n < KE - 1;
      if (t < KE - 1) {
        hok = hok && row >= 1;
        int hf;
        const long long hb = divmod(row >= 1 ? row - 1 : 0, nf, hf);
        hoff = (int)hb * (int)sb + hf * (int)sf + lane_c * isc + (nt + t - (KE - 1)) * ist;
      }
      asm volatile("" : "+v"(off), "+v"(hoff));                     // not hoisted as NC separate offsets
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        cur[ci] = x[(unsigned)(lane_c + ci < cin ? off + ci * isc : 0)];
        halo[ci] = 0.f;
      }
      if (hok) {
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) halo[ci] = x[(unsigned)(lane_c + ci < cin ? hoff + ci * isc : 0)];
      }
      dst = out + b * o_sb + t * o_st + f * o_sf + 4 * kq;
    };
    const long long stride = (long long)gridDim.x * 8;
    long long tile = (long long)blockIdx.x * 8 + w;
    if (tile < ntiles) fetch(tile);
    while (tile < ntiles) {
      float a[KP / 4];
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        // all shifts first, with every lane active; then the per-lane choice (the lanes of a row share kq, hence cok)
        const bool cok = lane_c + ci < cin;
        const float cv = cok ? cur[ci] : 0.f, hv = cok ? halo[ci] : 0.f;
        const float r1 = dpp_f<0x111>(cv), r2 = dpp_f<0x112>(cv), r3 = dpp_f<0x113>(cv), r4 = dpp_f<0x114>(cv);   // row_shr:1..4
        const float h1 = dpp_f<0x103>(hv), h2 = dpp_f<0x102>(hv), h3 = dpp_f<0x101>(hv);                           // row_shl:3..1
        a[KE * ci + 4] = cv;
        a[KE * ci + 3] = t >= 1 ? (n >= 1 ? r1 : h1) : 0.f;
        a[KE * ci + 2] = t >= 2 ? (n >= 2 ? r2 : h2) : 0.f;
        a[KE * ci + 1] = t >= 3 ? (n >= 3 ? r3 : h3) : 0.f;
        a[KE * ci + 0] = t >= 4 ? (n >= 4 ? r4 : hv) : 0.f;
      }
      v8bf_t ab[w_items<KP, true>()];
      if constexpr (BF) pack_operand<KP>(a, ab);                     // once for both halves of the outputs
      float* const dcur = dst;
      const bool okc = pok;
      const long long next = tile + stride;
      if (next < ntiles) fetch(next);
#pragma unroll
      for (int j0 = 0; j0 < H / 16; j0 += 3) {
        v4f_t acc[3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const float4 bv = *reinterpret_cast<const float4*>(ldsb + 16 * (j0 + jj) + 4 * kq);
          acc[jj] = v4f_t{bv.x, bv.y, bv.z, bv.w};
        }
        if constexpr (BF)
          mfma_tiles_packed<KP, 3>(ab, ldsw_lane, j0, acc);
        else
          mfma_tiles<KP, 3, false>(a, ldsw_lane, j0, acc);
        if (okc) {
#pragma unroll
          for (int jj = 0; jj < 3; ++jj)
            *reinterpret_cast<float4*>(dcur + 16 * (j0 + jj)) = make_float4(acc[jj][0], acc[jj][1], acc[jj][2], acc[jj][3]);
        }
      }
      tile = next;
    }
    return;
  }
  if constexpr (!PIPE)
  for (long long tile = (long long)blockIdx.x * 8 + w; tile < ntiles; tile += (long long)gridDim.x * 8) {
    const long long p = tile * 16 + n;
    const long long pc = p < npts ? p : npts - 1;
    int t, f;
    const long long row = divmod(pc, nt, t);                        // (b, f) row of the point
    const long long b = divmod(row, nf, f);
    // K index k = kq * KQ + i is (channel, tap) = (KQ/5 * kq + i / 5, i % 5): KQ is a multiple of the 5 taps, so the
    // split is compile-time per i and an element offset is  lane part (channel block, frame) + a wave-uniform term —
    // 32-bit; the generic form (a division and a 64-bit multiply-add per element) cost more cycles than the MFMAs.
    const float* xr = x + b * sb + f * sf;
    const int isc = (int)sc, ist = (int)st;
    int lane_off = (KP / 20) * kq * isc + t * ist;
    int lane_c = (KP / 20) * kq;
    const float* sr = state_in ? state_in + ((b * cin) * nf + f) * (KE - 1) : x;
    int lane_so = lane_c * nf * (KE - 1) + (KE - 1) + t;
    asm volatile("" : "+v"(lane_off), "+v"(lane_c), "+v"(lane_so));   // not hoisted as 40 separate offsets
    float a[KP / 4];
    // one straight-line run of loads per case (a per-element test of state_in turned every element into
    // branch - load - wait)
    auto build = [&](auto has_state) {
#pragma unroll
      for (int i = 0; i < KP / 4; ++i) {
        const int ci = i / KE, k = i % KE;                           // compile-time after unrolling
        const bool cok = lane_c + ci < cin;
        const bool tok = t + k - (KE - 1) >= 0;
        const float* src = xr + (cok && tok ? lane_off + ci * isc + (k - (KE - 1)) * ist : 0);
        if constexpr (decltype(has_state)::value)
          src = tok ? src : sr + (cok ? lane_so + ci * nf * (KE - 1) + (k - (KE - 1)) : 0);
        const float v = *src;
        a[i] = (cok && (tok || decltype(has_state)::value)) ? v : 0.f;
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (state_in)
      build(std::true_type{});
    else
      build(std::false_type{});
    float* dst = out + b * o_sb + t * o_st + f * o_sf + 4 * kq;
#pragma unroll
    for (int j0 = 0; j0 < H / 16; j0 += 3) {
      v4f_t acc[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + 16 * (j0 + jj) + 4 * kq);
        acc[jj] = v4f_t{bv.x, bv.y, bv.z, bv.w};
      }
      mfma_tiles<KP, 3, BF>(a, ldsw_lane, j0, acc);
      if (p < npts) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
          *reinterpret_cast<float4*>(dst + 16 * (j0 + jj)) = make_float4(acc[jj][0], acc[jj][1], acc[jj][2], acc[jj][3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Mamba, phase 2: dbl[s, t, 0:40] = x_proj(SiLU(causal_depthwise_conv4(xi)))  — (dt 6 | B 16 | C 16 | pad)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_mamba_xproj_kernel(const float* __restrict__ xz, int nt, long long npts, const float* __restrict__ conv_w,
                      const float* __restrict__ conv_b, const float* __restrict__ wxT,
                      const float* __restrict__ conv_state, float* __restrict__ dbl) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts) return;
  const int t = (int)(p % nt);
  const long long s = p / nt;
  float acc[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) acc[j] = 0.f;
  auto taps = [&](int e4, float4 (&v)[KC]) {
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int tt = t - (KC - 1) + k;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tt >= 0) {
        v[k] = *reinterpret_cast<const float4*>(xz + (s * nt + tt) * (2 * E) + e4 * 4);
      } else if (conv_state) {
        v[k] = *reinterpret_cast<const float4*>(conv_state + (s * (KC - 1) + (KC - 1) + tt) * E + e4 * 4);
      }
    }
  };
  float4 vn[KC];
  taps(0, vn);                                  // next block's taps in flight behind this block's 160 FMAs
  for (int e4 = 0; e4 < E / 4; ++e4) {
    float xi[KC][4];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      xi[k][0] = vn[k].x;
      xi[k][1] = vn[k].y;
      xi[k][2] = vn[k].z;
      xi[k][3] = vn[k].w;
    }
    if (e4 + 1 < E / 4) taps(e4 + 1, vn);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = e4 * 4 + i;
      float u = conv_b[e];
#pragma unroll
      for (int k = 0; k < KC; ++k) u = fmaf(conv_w[e * KC + k], xi[k][i], u);
      u = silu_f(u);
#pragma unroll
      for (int j = 0; j < XP; ++j) acc[j] = fmaf(wxT[e * XP + j], u, acc[j]);
    }
  }
  float4* dst = reinterpret_cast<float4*>(dbl + p * XP);
#pragma unroll
  for (int i = 0; i < XP / 4; ++i) dst[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
}

// ---------------------------------------------------------------------------------------------------------
// Mamba, phase 3: the selective scan.  One block (3 waves) per sequence, one thread per inner channel e:
//   u_t = SiLU(conv4(xi)),  dt_t = softplus(dt_proj(dbl_t[0:6])),  h = exp(dt A) h + dt B_t u,  y = C_t.h + D u,
//   y *= SiLU(z_t).  The 16 states, A[e, :], dt_proj row and the conv taps stay in registers; dbl_t is wave-uniform.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(E)
sn_mamba_scan_kernel(const float* __restrict__ xz, const float* __restrict__ dbl, int nt,
                     const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                     const float* __restrict__ wdt, const float* __restrict__ bdt, const float* __restrict__ a,
                     const float* __restrict__ dpar, float* conv_state, float* ssm_state, int carry,
                     float* ybuf, int have_u) {
  const long long s = blockIdx.x;
  const int e = threadIdx.x;
  // the states as pairs: the update of two states is three packed-fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32)
  // next to its two v_exp_f32
  v2f_t A[NST / 2], h[NST / 2];
  float wd[RK], cw[KC];
#pragma unroll
  for (int n = 0; n < NST; ++n) {
    // exp(dt A) = 2^(dt * A log2 e): the factor is folded into A once, and the step uses the bare v_exp_f32 (the
    // library exp adds a denormal-range rescue per call, 5 more operations on each of the 16 states of a step;
    // a decay below 1e-38 is zero either way)
    A[n >> 1][n & 1] = a[e * NST + n] * 1.44269504088896340736f;
    h[n >> 1][n & 1] = (carry && ssm_state) ? ssm_state[(s * E + e) * NST + n] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < RK; ++r) wd[r] = wdt[e * RK + r];
#pragma unroll
  for (int k = 0; k < KC; ++k) cw[k] = conv_w[e * KC + k];
  const float cb = conv_b[e], bd = bdt[e], dp = dpar[e];
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  if (carry && conv_state) {
    x0 = conv_state[(s * 3 + 0) * E + e];
    x1 = conv_state[(s * 3 + 1) * E + e];
    x2 = conv_state[(s * 3 + 2) * E + e];
  }
  const float* xrow = xz + s * nt * (2 * E) + e;
  const float* drow = dbl + s * nt * XP;
  float* yrow = ybuf + s * nt * E + e;
  // Everything a step reads from memory is requested one step ahead: the recurrence itself is ~150 VALU operations,
  // a step that first waits for its 38 wave-uniform dbl_t scalars (scalar loads, L2 latency) took 2 us.
  // have_u: ybuf holds u = SiLU(conv4(xi)) (sn_mamba_conv_kernel); the step reads u_t (one step ahead) before it
  // overwrites the slot with y_t, and the conv taps are only needed for the carried state at the end
  float xi_n = nt > 0 ? (have_u ? yrow[0] : xrow[0]) : 0.f, z_n = nt > 0 ? xrow[E] : 0.f;
  float row_n[RK + 2 * NST];
#pragma unroll
  for (int i = 0; i < RK + 2 * NST; ++i) row_n[i] = nt > 0 ? drow[i] : 0.f;
  for (int t = 0; t < nt; ++t) {
    const float xi = xi_n, z = z_n;
    float row[RK + 2 * NST];
#pragma unroll
    for (int i = 0; i < RK + 2 * NST; ++i) row[i] = row_n[i];
    if (t + 1 < nt) {                                    // next step's operands are in flight during this one
      xi_n = have_u ? yrow[(long long)(t + 1) * E] : xrow[(long long)(t + 1) * (2 * E)];
      z_n = xrow[(long long)(t + 1) * (2 * E) + E];
#pragma unroll
      for (int i = 0; i < RK + 2 * NST; ++i) row_n[i] = drow[(long long)(t + 1) * XP + i];   // wave-uniform: scalar loads
    }
    float u = xi;
    if (!have_u) {
      u = cb;
      u = fmaf(cw[0], x0, u);
      u = fmaf(cw[1], x1, u);
      u = fmaf(cw[2], x2, u);
      u = fmaf(cw[3], xi, u);
      u = silu_f(u);
      x0 = x1;
      x1 = x2;
      x2 = xi;
    }
    float dtv = bd;
#pragma unroll
    for (int r = 0; r < RK; ++r) dtv = fmaf(wd[r], row[r], dtv);
    // softplus.  log1pf() expands to ~100 double-float operations per step (a third of the recurrence); log1p(e) =
    // log(u) * e / (u - 1) with u = 1 + e (the rounding of u cancels in the quotient) is accurate to a few ulp on the
    // hardware log and costs 8
    float dt = dtv;
    if (dtv <= 20.f) {
      const float ex = __expf(dtv), u1 = 1.f + ex;
      dt = u1 == 1.f ? ex : __logf(u1) * __fdividef(ex, u1 - 1.f);
    }
    const float dtu = dt * u;
    const v2f_t dt2 = {dt, dt}, dtu2 = {dtu, dtu};
    v2f_t y2 = {0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NST / 2; ++n) {
      const v2f_t ex = dt2 * A[n];
      const v2f_t dA = {__builtin_amdgcn_exp2f(ex[0]), __builtin_amdgcn_exp2f(ex[1])};
      const v2f_t bt = {row[RK + 2 * n], row[RK + 2 * n + 1]};
      const v2f_t ct = {row[RK + NST + 2 * n], row[RK + NST + 2 * n + 1]};
      h[n] = __builtin_elementwise_fma(dA, h[n], dtu2 * bt);
      y2 = __builtin_elementwise_fma(h[n], ct, y2);
    }
    float y = y2[0] + y2[1];
    y = fmaf(dp, u, y) * silu_f(z);
    yrow[(long long)t * E] = y;
  }
  if (ssm_state) {
#pragma unroll
    for (int n = 0; n < NST; ++n) ssm_state[(s * E + e) * NST + n] = h[n >> 1][n & 1];
  }
  if (conv_state && have_u) {                            // the last three xi of the chunk (older ones: the carried state)
    float xs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tt = nt - 3 + j;
      xs[j] = tt >= 0 ? xrow[(long long)tt * (2 * E)] : (tt == -1 ? x2 : (tt == -2 ? x1 : x0));
    }
    x0 = xs[0];
    x1 = xs[1];
    x2 = xs[2];
  }
  if (conv_state) {
    conv_state[(s * 3 + 0) * E + e] = x0;
    conv_state[(s * 3 + 1) * E + e] = x1;
    conv_state[(s * 3 + 2) * E + e] = x2;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Mamba, phase 4: out = pool_T(x) + out_proj(pool_T(y))   (out_proj is linear, so the time pooling of
// IPDnet2.py:345-349 is applied to its operand).  blockIdx.y = output half (48 channels).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_mamba_out_kernel(const float* __restrict__ ybuf, fnssl_btf_view xv, int nt, int nt2, int nf, int tp,
                    long long nout, const float* __restrict__ woT, int residual, float* out, long long o_sb,
                    long long o_st, long long o_sf) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= nout) return;
  const int t2 = (int)(p % nt2);
  const long long s = p / nt2;
  const int f = (int)(s % nf);
  const long long b = s / nf;
  const int half = blockIdx.y;
  const float inv = 1.f / (float)tp;
  float acc[48];
#pragma unroll
  for (int o = 0; o < 48; ++o) acc[o] = 0.f;
  const float* yb = ybuf + (s * nt + (long long)t2 * tp) * E;
  auto ysum = [&](int e4) {
    float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < tp; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(yb + (long long)i * E + e4 * 4);
      yv.x += v.x;
      yv.y += v.y;
      yv.z += v.z;
      yv.w += v.w;
    }
    return yv;
  };
  float4 ynext = ysum(0);                       // the next block is requested before this block's 192 FMAs
  for (int e4 = 0; e4 < E / 4; ++e4) {
    const float4 yv = ynext;
    if (e4 + 1 < E / 4) ynext = ysum(e4 + 1);
    const float yy[4] = {yv.x * inv, yv.y * inv, yv.z * inv, yv.w * inv};
    const float* we = woT + (e4 * 4) * H + half * 48;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int o = 0; o < 48; ++o) acc[o] = fmaf(we[i * H + o], yy[i], acc[o]);
  }
  if (residual) {
    const float* xb = xv.p + b * xv.sb + (long long)t2 * tp * xv.st + f * xv.sf + half * 48;
#pragma unroll
    for (int o4 = 0; o4 < 12; ++o4) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = 0; i < tp; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (long long)i * xv.st + o4 * 4);
        r.x += v.x;
        r.y += v.y;
        r.z += v.z;
        r.w += v.w;
      }
      acc[4 * o4] = fmaf(r.x, inv, acc[4 * o4]);
      acc[4 * o4 + 1] = fmaf(r.y, inv, acc[4 * o4 + 1]);
      acc[4 * o4 + 2] = fmaf(r.z, inv, acc[4 * o4 + 2]);
      acc[4 * o4 + 3] = fmaf(r.w, inv, acc[4 * o4 + 3]);
    }
  }
  float4* dst = reinterpret_cast<float4*>(out + b * o_sb + t2 * o_st + f * o_sf + half * 48);
#pragma unroll
  for (int o4 = 0; o4 < 12; ++o4) dst[o4] = make_float4(acc[4 * o4], acc[4 * o4 + 1], acc[4 * o4 + 2], acc[4 * o4 + 3]);
}

// ---------------------------------------------------------------------------------------------------------
// Head: FreqInverse (1x1 conv 96 -> 16 x 16 per compressed bin, scattered to the 16 fine bins it covers), tanh,
// decoder Linear(16, 16) and the output re-ordering of IPDnet2.py:355-364:
//   out[b, t, 2 f + gg, m, a] = dec[b, f, t, a*8 + gg*4 + m]
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sn_head_kernel(fnssl_btf_view xv, int nt2, int nfc, long long npts, const float* __restrict__ wfiP,
               const float* __restrict__ bfiP, const float* __restrict__ wdT, const float* __restrict__ bd,
               float* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts) return;
  const int fc = (int)(p % nfc);
  const int t2 = (int)((p / nfc) % nt2);
  const long long b = p / ((long long)nfc * nt2);
  float x[H];
  load_row(x, xv.p + b * xv.sb + t2 * xv.st + fc * xv.sf);
  const int nf = nfc * 16;
  float* orow = out + ((b * nt2 + t2) * (2LL * nf) + 2LL * fc * 16) * 8;
#pragma unroll 1
  for (int r = 0; r < 16; ++r) {
    float dec[DO];
#pragma unroll
    for (int j = 0; j < DO; ++j) dec[j] = bd[j];
#pragma unroll 1
    for (int o = 0; o < DO; ++o) {                 // rolled: 96 + 16 scalar weights per trip
      const float* wr = wfiP + (r * DO + o) * H;
      float a0 = bfiP[r * DO + o], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int h = 0; h < H; h += 4) {
        a0 = fmaf(wr[h], x[h], a0);
        a1 = fmaf(wr[h + 1], x[h + 1], a1);
        a2 = fmaf(wr[h + 2], x[h + 2], a2);
        a3 = fmaf(wr[h + 3], x[h + 3], a3);
      }
      const float v = tanhf((a0 + a1) + (a2 + a3));
#pragma unroll
      for (int j = 0; j < DO; ++j) dec[j] = fmaf(wdT[o * DO + j], v, dec[j]);
    }
    float4* dst = reinterpret_cast<float4*>(orow + r * 16);
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {          // element gg*8 + m*2 + a  <-  dec[a*8 + gg*4 + m]
      dst[gg * 2] = make_float4(dec[gg * 4], dec[8 + gg * 4], dec[gg * 4 + 1], dec[8 + gg * 4 + 1]);
      dst[gg * 2 + 1] = make_float4(dec[gg * 4 + 2], dec[8 + gg * 4 + 2], dec[gg * 4 + 3], dec[8 + gg * 4 + 3]);
    }
  }
}

// Head on the matrix pipe.  Fine bin r of a compressed bin is output tile r of FreqInverse's 96 -> 256 product; its
// D fragment (lane (point, q): outputs 4 q .. + 3) after bias + tanh is exactly the B operand of the decoder's
// 16 -> 16 product (K = 16: lane q holds k = 4 q .. + 3), so the decoder is one more group of 4 MFMAs per fine bin.
// Output order of IPDnet2.py:355-364: element gg*8 + m*2 + a <- dec[a*8 + gg*4 + m]: lane q holds (a, gg) = (q >> 1,
// q & 1), m = 0..3; it trades registers with lane q ^ 2 (the other a) and stores one float4.
__device__ __forceinline__ float tanh_fast(float x) {       // 1 - 2 / (e^2x + 1): the LSTM kernels' formulation
  return __fmaf_rn(-2.0f, __builtin_amdgcn_rcpf(__fadd_rn(__expf(2.0f * x), 1.0f)), 1.0f);
}

__global__ void __launch_bounds__(512)
sn_head_mfma_kernel(fnssl_btf_view xv, int nt2, int nfc, long long npts, const float* __restrict__ wfiP,
                    const float* __restrict__ bfiP, const float* __restrict__ wdT, const float* __restrict__ bd,
                    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float ldsw[];     // FreqInverse: 16 tiles x 6 groups x 1 KiB; decoder: 1 KiB
  float* ldsd = ldsw + 16 * 6 * 256;
  fill_w_lds_strided<H, 16 * DO, 512>(ldsw, wfiP, 1, H, H, 16 * DO);
  fill_w_lds<DO, DO, 512>(ldsd, wdT, DO, DO, DO);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
  const float* ldsw_lane = ldsw + lane * 4;
  const float* ldsd_lane = ldsd + lane * 4;
  const long long ntiles = (npts + 15) / 16;
  const float4 bdv = *reinterpret_cast<const float4*>(bd + 4 * q);
  for (long long tile = (long long)blockIdx.x * 8 + w; tile < ntiles; tile += (long long)gridDim.x * 8) {
    const long long p = tile * 16 + n;
    const long long pc = p < npts ? p : npts - 1;
    int fc, t2;
    const long long b = divmod(divmod(pc, nfc, fc), nt2, t2);
    const float4* row = reinterpret_cast<const float4*>(xv.p + b * xv.sb + t2 * xv.st + fc * xv.sf + q * (H / 4));
    float a[H / 4];
#pragma unroll
    for (int i = 0; i < H / 16; ++i) {
      const float4 v = row[i];
      a[4 * i] = v.x;
      a[4 * i + 1] = v.y;
      a[4 * i + 2] = v.z;
      a[4 * i + 3] = v.w;
    }
    const int nf = nfc * 16;
    float* orow = out + ((b * nt2 + t2) * (2LL * nf) + 2LL * fc * 16) * 8 + (q & 1) * 8 + (q >> 1) * 4;
#pragma unroll 1
    for (int j0 = 0; j0 < 16; j0 += 4) {
      v4f_t acc[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float4 bv = *reinterpret_cast<const float4*>(bfiP + (j0 + jj) * DO + 4 * q);
        acc[jj] = v4f_t{bv.x, bv.y, bv.z, bv.w};
      }
      mfma_tiles<H, 4>(a, ldsw_lane, j0, acc);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float v[4] = {tanh_fast(acc[jj][0]), tanh_fast(acc[jj][1]), tanh_fast(acc[jj][2]), tanh_fast(acc[jj][3])};
        v4f_t dec[1] = {v4f_t{bdv.x, bdv.y, bdv.z, bdv.w}};
        mfma_tiles<DO, 1>(v, ldsd_lane, 0, dec);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __shfl_xor(dec[0][i], 32, 64);
        const float4 st = (q >> 1) == 0 ? make_float4(dec[0][0], o[0], dec[0][1], o[1])
                                        : make_float4(o[2], dec[0][2], o[3], dec[0][3]);
        if (p < npts) *reinterpret_cast<float4*>(orow + (j0 + jj) * 16) = st;
      }
    }
  }
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline bool view_ok(const fnssl_btf_view* v) {
  return v && v->p && (reinterpret_cast<size_t>(v->p) % 16 == 0) && v->sb % 4 == 0 && v->st % 4 == 0 && v->sf % 4 == 0;
}
inline bool out_ok(const float* p, long long sb, long long st, long long sf) {
  return p && (reinterpret_cast<size_t>(p) % 16 == 0) && sb % 4 == 0 && st % 4 == 0 && sf % 4 == 0;
}
inline unsigned blocks_of(long long n) { return (unsigned)((n + 255) / 256); }

struct MambaWs {
  float *xz, *dbl, *y;
};
inline size_t mamba_ws_floats(long long npts) { return (size_t)npts * (2 * E + XP + E); }
inline MambaWs carve_mamba(float* ws, long long npts) {
  MambaWs m;
  m.xz = ws;
  m.dbl = ws + npts * (2 * E);
  m.y = m.dbl + npts * XP;
  return m;
}

}  // namespace

// workgroups of the matrix-pipe kernels: 8 waves x one 16-point tile each per pass, per_cu workgroups per CU (what
// the kernel's LDS weight image allows)
static bool precision_ok(int p) { return p == FNSSL_PRECISION_FP32 || p == FNSSL_PRECISION_BF16; }

static unsigned mfma_grid(long long npts, int per_cu, int waves = 8) {
  const long long wgs = ((npts + 15) / 16 + waves - 1) / waves;
  const long long cap = (long long)fnssl::device_cus() * per_cu;
  return (unsigned)(wgs < cap ? wgs : cap);
}

extern "C" {

int fnssl_sn_layernorm(const float* x, long long rows, int h, const float* w, const float* b, float eps, float* y,
                       void* stream) {
  FNSSL_REQUIRE(x && w && b && y, "sn_layernorm: null pointer");
  FNSSL_REQUIRE(rows >= 0 && h > 0 && h <= 1024, "sn_layernorm: bad shape (rows %lld, h %d)", rows, h);
  if (rows == 0) return FNSSL_OK;
  FNSSL_REQUIRE((rows + 3) / 4 < (1ll << 31), "sn_layernorm: too many rows");
  fnssl::TimedLaunch tl("sn_layernorm", fnssl::as_stream(stream));
  hipLaunchKernelGGL(sn_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, fnssl::as_stream(stream), x,
                     rows, h, w, b, eps, y);
  FNSSL_CHECK_LAUNCH("sn_layernorm_kernel");
  return FNSSL_OK;
}

int fnssl_sn_encoder(const float* x, long long x_sb, long long x_sc, long long x_sf, long long x_st, int nb, int cin,
                     int nf, int nt, const float* wT, const float* bias, const float* state_in, float* state_out,
                     float* out, long long o_sb, long long o_st, long long o_sf, int precision, void* stream) {
  FNSSL_REQUIRE(x && wT && bias, "sn_encoder: null pointer");
  FNSSL_REQUIRE(precision_ok(precision), "sn_encoder: precision %d (FNSSL_PRECISION_FP32 or _BF16)", precision);
  FNSSL_REQUIRE(nb > 0 && cin > 0 && nf > 0 && nt > 0, "sn_encoder: empty problem");
  FNSSL_REQUIRE(out_ok(out, o_sb, o_st, o_sf), "sn_encoder: output must be 16-byte aligned with strides %% 4 == 0");
  const long long npts = (long long)nb * nf * nt;
  FNSSL_REQUIRE(blocks_of(npts) < (1u << 31), "sn_encoder: too many points");
  hipStream_t s = fnssl::as_stream(stream);
  {
    fnssl::TimedLaunch tl("sn_encoder", s, 2.0 * npts * cin * KE * H);
    const int kk = cin * KE;
    // the matrix-pipe kernel addresses a (b, f) row's channels and frames with 32-bit element offsets
    const bool off32 = (long double)cin * (x_sc < 0 ? -x_sc : x_sc) + (long double)nt * (x_st < 0 ? -x_st : x_st) < 2.0e9L &&
                       (long double)cin * nf * (KE - 1) + nt < 2.0e9L;
    // the whole-signal kernel addresses every element as a non-negative 32-bit offset from x
    const bool pos32 = x_sb >= 0 && x_sc >= 0 && x_sf >= 0 && x_st >= 0 &&
                       (long double)nb * x_sb + (long double)cin * x_sc + (long double)nf * x_sf + (long double)nt * x_st < 2.0e9L;
    const bool bf = precision == FNSSL_PRECISION_BF16;
    FNSSL_REQUIRE(!bf || (kk <= 160 && off32), "sn_encoder: FNSSL_PRECISION_BF16 needs cin * 5 <= 160 (cin %d) and 32-bit offsets",
                  cin);
    if ((bf || !fnssl::tune(FNSSL_TUNE_SN_SCALAR)) && kk <= 160 && off32) {
#define FNSSL_SN_ENC_P(KP, BF, PIPE)                                                                                \
  do {                                                                                                              \
    const size_t lds = (size_t)(w_lds_floats<KP, H, BF>() + H) * sizeof(float);                                     \
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_encoder_mfma_kernel<KP, BF, PIPE>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
    hipLaunchKernelGGL((sn_encoder_mfma_kernel<KP, BF, PIPE>), dim3(mfma_grid(npts, 2)), dim3(512), lds, s, x, x_sb, \
                       x_sc, x_sf, x_st, cin, nf, nt, npts, wT, bias, state_in, out, o_sb, o_st, o_sf);             \
  } while (0)
#define FNSSL_SN_ENC(KP, BF)                                                                                        \
  do {                                                                                                              \
    if (!state_in && nt >= KE - 1 && pos32) FNSSL_SN_ENC_P(KP, BF, true); else FNSSL_SN_ENC_P(KP, BF, false);       \
  } while (0)
      if (kk <= 80) {
        if (bf) FNSSL_SN_ENC(80, true); else FNSSL_SN_ENC(80, false);
      } else {
        if (bf) FNSSL_SN_ENC(160, true); else FNSSL_SN_ENC(160, false);
      }
#undef FNSSL_SN_ENC_P
#undef FNSSL_SN_ENC
    } else {
      hipLaunchKernelGGL(sn_encoder_kernel, dim3(blocks_of(npts)), dim3(256), 0, s, x, x_sb, x_sc, x_sf, x_st, cin, nf, nt,
                         npts, wT, bias, state_in, out, o_sb, o_st, o_sf);
    }
    FNSSL_CHECK_LAUNCH("sn_encoder_kernel");
  }
  if (state_out) {
    const long long nrows = (long long)nb * cin * nf;
    hipLaunchKernelGGL(sn_encoder_state_kernel, dim3(blocks_of(nrows)), dim3(256), 0, s, x, x_sb, x_sc, x_sf, x_st, cin,
                       nf, nt, nrows, state_in, state_out);
    FNSSL_CHECK_LAUNCH("sn_encoder_state_kernel");
  }
  return FNSSL_OK;
}

int fnssl_sn_fconv(const fnssl_btf_view* x, int nb, int nt, int nf, const fnssl_sn_fconv_w* w, int residual, int pool,
                   float* out, long long o_sb, long long o_st, long long o_sf, int precision, void* stream) {
  FNSSL_REQUIRE(view_ok(x), "sn_fconv: x must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(precision_ok(precision), "sn_fconv: precision %d (FNSSL_PRECISION_FP32 or _BF16)", precision);
  FNSSL_REQUIRE(w && w->ln_w && w->ln_b && w->wT && w->bias && w->prelu, "sn_fconv: null weights");
  FNSSL_REQUIRE(out_ok(out, o_sb, o_st, o_sf), "sn_fconv: output must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(nb > 0 && nt > 0, "sn_fconv: empty problem");
  FNSSL_REQUIRE(pow2(nf) && nf >= 8 && nf <= 256, "sn_fconv: nf must be a power of two in [8, 256], got %d", nf);
  FNSSL_REQUIRE(pool == 1 || pool == 2 || pool == 8, "sn_fconv: pool must be 1, 2 or 8");
  const long long nframes = (long long)nb * nt;
  const int lg = ilog2(nf);
  const long long nblk = (nframes + (256 >> lg) - 1) / (256 >> lg);
  FNSSL_REQUIRE(nblk < (1ll << 31), "sn_fconv: too many frames");
  hipStream_t s = fnssl::as_stream(stream);
  fnssl::TimedLaunch tl(nf > 128 ? "sn_fconv_f256" : (nf > 16 ? "sn_fconv_f128" : "sn_fconv_f16"), s,
                        2.0 * nframes * nf * H * CG * KF);
  // below 16 bins there is no matrix-pipe kernel: the product stays exact fp32 in both modes (include/fnssl.h)
  const bool bf = precision == FNSSL_PRECISION_BF16 && nf >= 16;
  const bool mfma = (bf || !fnssl::tune(FNSSL_TUNE_SN_SCALAR)) && nf >= 16;
  const size_t lds = (size_t)(NG * (bf ? w_lds_floats<64, 16, true>() : w_lds_floats<64, 16, false>()) + 4 * H +
                              (256 + 4 * (256 >> lg)) * 100) * sizeof(float);
  const long long cus = fnssl::device_cus();
#define FNSSL_SN_FCONV_M(P, BF)                                                                                   \
  do {                                                                                                            \
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_fconv_mfma_kernel<P, BF>),                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                      \
    hipLaunchKernelGGL((sn_fconv_mfma_kernel<P, BF>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(1024), lds, s, *x, \
                       nt, nf, lg, nframes, nblk, *w, residual, out, o_sb, o_st, o_sf);                           \
  } while (0)
#define FNSSL_SN_FCONV(P)                                                                                         \
  if (mfma) {                                                                                                     \
    if (bf) FNSSL_SN_FCONV_M(P, true); else FNSSL_SN_FCONV_M(P, false);                                           \
  } else                                                                                                          \
    hipLaunchKernelGGL(sn_fconv_kernel<P>, dim3((unsigned)nblk), dim3(256), 0, s, *x, nt, nf, lg, nframes, *w, residual, \
                       out, o_sb, o_st, o_sf)
  if (pool == 1) {
    FNSSL_SN_FCONV(1);
  } else if (pool == 2) {
    FNSSL_SN_FCONV(2);
  } else {
    FNSSL_SN_FCONV(8);
  }
#undef FNSSL_SN_FCONV
#undef FNSSL_SN_FCONV_M
  FNSSL_CHECK_LAUNCH("sn_fconv_kernel");
  return FNSSL_OK;
}

int fnssl_sn_full(const fnssl_btf_view* x, int nb, int nt, int nf, const fnssl_sn_full_w* w, int residual, float* out,
                  long long o_sb, long long o_st, long long o_sf, int precision, void* stream) {
  FNSSL_REQUIRE(view_ok(x), "sn_full: x must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(precision_ok(precision), "sn_full: precision %d (FNSSL_PRECISION_FP32 or _BF16)", precision);
  FNSSL_REQUIRE(w && w->ln_w && w->ln_b && w->wsT && w->bs && w->wfT && w->bf && w->wuT && w->bu, "sn_full: null weights");
  FNSSL_REQUIRE(out_ok(out, o_sb, o_st, o_sf), "sn_full: output must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(nb > 0 && nt > 0, "sn_full: empty problem");
  FNSSL_REQUIRE(pow2(nf) && nf >= 8 && nf <= 256, "sn_full: nf must be a power of two in [8, 256], got %d", nf);
  const long long nframes = (long long)nb * nt;
  const int lg = ilog2(nf);
  const long long nblk = (nframes + (256 >> lg) - 1) / (256 >> lg);
  FNSSL_REQUIRE(nblk < (1ll << 31), "sn_full: too many frames");
  hipStream_t s = fnssl::as_stream(stream);
  fnssl::TimedLaunch tl("sn_full", s, 2.0 * nframes * nf * (2.0 * H * HS + (double)HS * nf));
  // the matrix-pipe kernel exists for 16, 64 and 128 bins; elsewhere the products stay exact fp32 in both modes
  const bool bf = precision == FNSSL_PRECISION_BF16 && (nf == 16 || nf == 64 || nf == 128);
  const bool mfma = (bf || !fnssl::tune(FNSSL_TUNE_SN_SCALAR)) && (nf == 16 || nf == 64 || nf == 128);
  if (mfma) {
    const long long cus = fnssl::device_cus();
    const unsigned grid = (unsigned)(nblk < cus ? nblk : cus);
    const size_t lds = (size_t)((nf / 16) * (nf / 16) * 256 + 16 * 256 + 3 * H + 16 + nf + 2 * 256 * 8) * sizeof(float);
#define FNSSL_SN_FULL_P(NFV, BF)                                                                                   \
  do {                                                                                                             \
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_full_mfma_kernel<NFV, BF>),                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                       \
    hipLaunchKernelGGL((sn_full_mfma_kernel<NFV, BF>), dim3(grid), dim3(1024), lds, s, *x, nt, nframes, nblk, *w,  \
                       residual, out, o_sb, o_st, o_sf);                                                           \
  } while (0)
#define FNSSL_SN_FULL(NFV)                                                                                         \
  do {                                                                                                             \
    if (bf) FNSSL_SN_FULL_P(NFV, true); else FNSSL_SN_FULL_P(NFV, false);                                          \
  } while (0)
    if (nf == 16)
      FNSSL_SN_FULL(16);
    else if (nf == 64)
      FNSSL_SN_FULL(64);
    else
      FNSSL_SN_FULL(128);
#undef FNSSL_SN_FULL
#undef FNSSL_SN_FULL_P
  } else {
    hipLaunchKernelGGL(sn_full_kernel, dim3((unsigned)nblk), dim3(256), 0, s, *x, nt, nf, lg, nframes, *w, residual, out,
                       o_sb, o_st, o_sf);
  }
  FNSSL_CHECK_LAUNCH("sn_full_kernel");
  return FNSSL_OK;
}

size_t fnssl_sn_mamba_workspace_bytes(int nb, int nt, int nf) {
  if (nb <= 0 || nt <= 0 || nf <= 0) return 0;
  return mamba_ws_floats((long long)nb * nt * nf) * sizeof(float);
}

int fnssl_sn_mamba(const fnssl_btf_view* x, int nb, int nt, int nf, const fnssl_sn_mamba_w* w, int residual,
                   int time_pool, float* conv_state, float* ssm_state, int carry, float* out, long long o_sb,
                   long long o_st, long long o_sf, void* workspace, size_t workspace_bytes, int precision, void* stream) {
  FNSSL_REQUIRE(view_ok(x), "sn_mamba: x must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(precision_ok(precision), "sn_mamba: precision %d (FNSSL_PRECISION_FP32 or _BF16)", precision);
  FNSSL_REQUIRE(w && w->ln_w && w->ln_b && w->winT && w->conv_w && w->conv_b && w->wxT && w->wdt && w->bdt && w->a &&
                    w->d && w->woT, "sn_mamba: null weights");
  FNSSL_REQUIRE(out_ok(out, o_sb, o_st, o_sf), "sn_mamba: output must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(nb > 0 && nt > 0 && nf > 0, "sn_mamba: empty problem");
  FNSSL_REQUIRE(time_pool >= 1 && time_pool <= 16, "sn_mamba: time_pool must be in [1, 16]");
  FNSSL_REQUIRE(!carry || (conv_state && ssm_state), "sn_mamba: carry needs both state buffers");
  const long long npts = (long long)nb * nt * nf, nseq = (long long)nb * nf;
  if (workspace_bytes < fnssl_sn_mamba_workspace_bytes(nb, nt, nf) || !workspace) {
    fnssl::set_error("sn_mamba: workspace %zu < %zu bytes", workspace_bytes, fnssl_sn_mamba_workspace_bytes(nb, nt, nf));
    return FNSSL_E_WORKSPACE;
  }
  FNSSL_REQUIRE(blocks_of(npts) < (1u << 31) && nseq < (1ll << 31), "sn_mamba: too many points");
  const MambaWs m = carve_mamba(static_cast<float*>(workspace), npts);
  hipStream_t s = fnssl::as_stream(stream);
  const bool bf = precision == FNSSL_PRECISION_BF16;
  const bool mfma = bf || !fnssl::tune(FNSSL_TUNE_SN_SCALAR);   // A/B (fp32 only): the scalar-operand kernels
  {
    fnssl::TimedLaunch tl("sn_mamba_in", s, 2.0 * npts * H * 2 * E);
    if (mfma) {
      const unsigned nwg = mfma_grid(npts, 1, 16);
      if (bf) {
        const size_t lds = (size_t)w_lds_floats<H, 2 * E, true>() * sizeof(float);
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_mamba_in_mfma_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sn_mamba_in_mfma_kernel<true>, dim3(nwg), dim3(1024), lds, s, *x, nt, nf, npts, w->ln_w,
                           w->ln_b, w->winT, m.xz);
      } else {
        const size_t lds = (size_t)w_lds_floats<H, 2 * E, false>() * sizeof(float);
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_mamba_in_mfma_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sn_mamba_in_mfma_kernel<false>, dim3(nwg), dim3(1024), lds, s, *x, nt, nf, npts, w->ln_w,
                           w->ln_b, w->winT, m.xz);
      }
    } else {
      hipLaunchKernelGGL(sn_mamba_in_kernel, dim3(blocks_of(npts), 4), dim3(256), 0, s, *x, nt, nf, npts, w->ln_w, w->ln_b,
                         w->winT, m.xz);
    }
    FNSSL_CHECK_LAUNCH("sn_mamba_in_kernel");
  }
  {
    fnssl::TimedLaunch tl("sn_mamba_xproj", s, 2.0 * npts * E * (XP - 2 + KC));
    if (mfma && !carry) {                                 // whole sequences: conv + x_proj in one pass, u not materialised
      const long long tiles = nseq * ((nt + 12) / 13);
      if (bf) {
        constexpr size_t lds = (size_t)(w_lds_floats<E, 48, true>() + E * KC + E) * sizeof(float);
        hipLaunchKernelGGL(sn_mamba_convx_mfma_kernel<true>, dim3(mfma_grid(tiles * 16, 4)), dim3(512), lds, s, m.xz, nt, nseq,
                           w->conv_w, w->conv_b, w->wxT, m.dbl);
      } else {
        constexpr size_t lds = (size_t)(w_lds_floats<E, 48, false>() + E * KC + E) * sizeof(float);
        hipLaunchKernelGGL(sn_mamba_convx_mfma_kernel<false>, dim3(mfma_grid(tiles * 16, 4)), dim3(512), lds, s, m.xz, nt, nseq,
                           w->conv_w, w->conv_b, w->wxT, m.dbl);
      }
    } else if (mfma) {
      const long long nq = npts * (E / 4);
      FNSSL_REQUIRE(blocks_of(nq) < (1u << 31), "sn_mamba: too many points");
      hipLaunchKernelGGL(sn_mamba_conv_kernel, dim3(blocks_of(nq)), dim3(256), 0, s, m.xz, nt, npts, w->conv_w, w->conv_b,
                         carry ? conv_state : nullptr, m.y);
      FNSSL_CHECK_LAUNCH("sn_mamba_conv_kernel");
      constexpr size_t lds_bf = (size_t)w_lds_floats<E, 48, true>() * sizeof(float);
      constexpr size_t lds_f32 = (size_t)w_lds_floats<E, 48, false>() * sizeof(float);
      if (bf)
        hipLaunchKernelGGL(sn_mamba_xproj_mfma_kernel<true>, dim3(mfma_grid(npts, 4)), dim3(512), lds_bf, s, m.y, npts,
                           w->wxT, m.dbl);
      else
        hipLaunchKernelGGL(sn_mamba_xproj_mfma_kernel<false>, dim3(mfma_grid(npts, 4)), dim3(512), lds_f32, s, m.y, npts,
                           w->wxT, m.dbl);
    } else {
      hipLaunchKernelGGL(sn_mamba_xproj_kernel, dim3(blocks_of(npts)), dim3(256), 0, s, m.xz, nt, npts, w->conv_w, w->conv_b,
                         w->wxT, carry ? conv_state : nullptr, m.dbl);
    }
    FNSSL_CHECK_LAUNCH("sn_mamba_xproj_kernel");
  }
  {
    fnssl::TimedLaunch tl("sn_mamba_scan", s, (double)npts * E * (7.0 * NST + 2 * RK + 2 * KC));
    hipLaunchKernelGGL(sn_mamba_scan_kernel, dim3((unsigned)nseq), dim3(E), 0, s, m.xz, m.dbl, nt, w->conv_w, w->conv_b,
                       w->wdt, w->bdt, w->a, w->d, conv_state, ssm_state, carry, m.y, mfma && carry ? 1 : 0);
    FNSSL_CHECK_LAUNCH("sn_mamba_scan_kernel");
  }
  {
    const int nt2 = nt / time_pool;
    const long long nout = nseq * nt2;
    if (nout > 0) {
      fnssl::TimedLaunch tl("sn_mamba_out", s, 2.0 * nout * E * H);
      if (mfma) {
#define FNSSL_SN_OUT(BF, TP)                                                                                        \
  do {                                                                                                              \
    const size_t lds = (size_t)w_lds_floats<E, H, BF>() * sizeof(float);                                            \
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_mamba_out_mfma_kernel<BF, TP>),                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
    hipLaunchKernelGGL((sn_mamba_out_mfma_kernel<BF, TP>), dim3(mfma_grid(nout, 2)), dim3(512), lds, s, m.y, *x, nt, \
                       nt2, nf, time_pool, nout, w->woT, residual, out, o_sb, o_st, o_sf);                          \
  } while (0)
        if (bf) {
          if (time_pool == 1) FNSSL_SN_OUT(true, 1); else if (time_pool == 5) FNSSL_SN_OUT(true, 5); else FNSSL_SN_OUT(true, 0);
        } else {
          if (time_pool == 1) FNSSL_SN_OUT(false, 1); else if (time_pool == 5) FNSSL_SN_OUT(false, 5); else FNSSL_SN_OUT(false, 0);
        }
#undef FNSSL_SN_OUT
      } else {
        hipLaunchKernelGGL(sn_mamba_out_kernel, dim3(blocks_of(nout), 2), dim3(256), 0, s, m.y, *x, nt, nt2, nf, time_pool,
                           nout, w->woT, residual, out, o_sb, o_st, o_sf);
      }
      FNSSL_CHECK_LAUNCH("sn_mamba_out_kernel");
    }
  }
  return FNSSL_OK;
}

int fnssl_sn_head(const fnssl_btf_view* x, int nb, int nt2, int nfc, const float* wfiP, const float* bfiP,
                  const float* wdT, const float* bd, float* out, void* stream) {
  FNSSL_REQUIRE(view_ok(x), "sn_head: x must be 16-byte aligned with strides %% 4 == 0");
  FNSSL_REQUIRE(wfiP && bfiP && wdT && bd && out && reinterpret_cast<size_t>(out) % 16 == 0, "sn_head: null / unaligned pointer");
  FNSSL_REQUIRE(nb > 0 && nfc > 0 && nt2 >= 0, "sn_head: empty problem");
  if (nt2 == 0) return FNSSL_OK;
  const long long npts = (long long)nb * nt2 * nfc;
  hipStream_t s = fnssl::as_stream(stream);
  fnssl::TimedLaunch tl("sn_head", s, 2.0 * npts * 16 * (DO * H + DO * DO));
  if (!fnssl::tune(FNSSL_TUNE_SN_SCALAR)) {
    const size_t lds = (size_t)(16 * 6 * 256 + 256) * sizeof(float);
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_head_mfma_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sn_head_mfma_kernel, dim3(mfma_grid(npts, 1)), dim3(512), lds, s, *x, nt2, nfc, npts, wfiP, bfiP, wdT,
                       bd, out);
  } else {
    hipLaunchKernelGGL(sn_head_kernel, dim3(blocks_of(npts)), dim3(256), 0, s, *x, nt2, nfc, npts, wfiP, bfiP, wdT, bd, out);
  }
  FNSSL_CHECK_LAUNCH("sn_head_kernel");
  return FNSSL_OK;
}

// ---- whole network ------------------------------------------------------------------------------------------

static size_t sn_act_floats(int nb, int nf, int nt, int ratio) {
  const size_t fr = (size_t)nb * nt;
  const int nfc = nf / 16, nt2 = nt / ratio;
  return fr * nf * H + fr * (nf / 2) * H + fr * nfc * H + (size_t)nb * (nt2 > 0 ? nt2 : 1) * nfc * H;
}

size_t fnssl_sn_forward_workspace_bytes(int nb, int nf, int nt) {
  if (nb <= 0 || nf <= 0 || nt <= 0) return 0;
  // sized for ANY time ratio fnssl_sn_forward accepts (1..16): the pooled buffer is largest at ratio 1
  return (sn_act_floats(nb, nf, nt, 1) + mamba_ws_floats((long long)nb * nt * (nf / 16))) * sizeof(float) + 256;
}

size_t fnssl_sn_state_floats(const fnssl_sn_net* net, int nb, int nf) {
  if (!net || nb <= 0 || nf <= 0) return 0;
  const size_t nseq = (size_t)nb * (nf / 16);
  return (size_t)nb * net->dim_input * nf * (KE - 1) + (size_t)net->num_layers * 2 * nseq * (3 * E + E * NST);
}

int fnssl_sn_forward(const fnssl_sn_net* net, const float* x, long long x_sb, long long x_sc, long long x_sf,
                     long long x_st, int nb, int nf, int nt, float* state, int carry, float* out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  FNSSL_REQUIRE(net && x && out, "sn_forward: null pointer");
  FNSSL_REQUIRE(net->num_layers >= 1 && net->num_layers <= FNSSL_SN_MAX_LAYERS, "sn_forward: 1..%d layers", FNSSL_SN_MAX_LAYERS);
  FNSSL_REQUIRE(net->time_ratio >= 1 && net->time_ratio <= 16, "sn_forward: time_ratio in [1, 16]");
  FNSSL_REQUIRE(nb > 0 && nt > 0, "sn_forward: empty problem");
  FNSSL_REQUIRE(precision_ok(net->precision), "sn_forward: net->precision %d (FNSSL_PRECISION_FP32 or _BF16)", net->precision);
  const int prec = net->precision;
  FNSSL_REQUIRE(nf == 128 || nf == 256, "sn_forward: num_freqs must be 128 or 256 (got %d)", nf);
  FNSSL_REQUIRE(!carry || state, "sn_forward: carry needs the state buffer");
  FNSSL_REQUIRE(!state || nt % net->time_ratio == 0, "sn_forward: streaming chunks must be multiples of %d frames", net->time_ratio);
  const int ratio = net->time_ratio, nfc = nf / 16, nt2 = nt / ratio;
  const size_t need = (sn_act_floats(nb, nf, nt, ratio) + mamba_ws_floats((long long)nb * nt * nfc)) * sizeof(float) + 256;
  if (!workspace || workspace_bytes < need) {
    fnssl::set_error("sn_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    return FNSSL_E_WORKSPACE;
  }
  float* base = reinterpret_cast<float*>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
  const size_t fr = (size_t)nb * nt;
  float* a0 = base;                              // [nb, nt, nf, H]
  float* a1 = a0 + fr * nf * H;                  // [nb, nt, nf/2, H]
  float* a2 = a1 + fr * (nf / 2) * H;            // [nb, nt, nfc, H]
  float* a3 = a2 + fr * nfc * H;                 // [nb, nt2, nfc, H]
  float* mws = a3 + (size_t)nb * (nt2 > 0 ? nt2 : 1) * nfc * H;
  const size_t mws_bytes = mamba_ws_floats((long long)nb * nt * nfc) * sizeof(float);
  const size_t nseq = (size_t)nb * nfc;
  float* st_enc = state;
  float* st_m = state ? state + (size_t)nb * net->dim_input * nf * (KE - 1) : nullptr;
  auto conv_st = [&](int l, int j) { return st_m ? st_m + ((size_t)l * 2 + j) * nseq * (3 * E + E * NST) : nullptr; };
  auto ssm_st = [&](int l, int j) { return st_m ? conv_st(l, j) + nseq * 3 * E : nullptr; };
  int rc;
#define FNSSL_SN_TRY(call) \
  do {                     \
    rc = (call);           \
    if (rc != FNSSL_OK) return rc; \
  } while (0)
  auto strides = [&](int f, long long& sb, long long& st, long long& sf, int t) {
    sf = H;
    st = (long long)f * H;
    sb = (long long)t * f * H;
  };
  long long sb, st, sf, sb2, st2, sf2;
  strides(nf, sb, st, sf, nt);
  FNSSL_SN_TRY(fnssl_sn_encoder(x, x_sb, x_sc, x_sf, x_st, nb, net->dim_input, nf, nt, net->enc_wT, net->enc_b,
                                carry ? st_enc : nullptr, st_enc, a0, sb, st, sf, prec, stream));
  const fnssl_sn_layer* L = &net->layers[0];
  fnssl_btf_view v0 = {a0, sb, st, sf};
  strides(nf / 2, sb2, st2, sf2, nt);
  FNSSL_SN_TRY(fnssl_sn_fconv(&v0, nb, nt, nf, &L->fconv1, 1, 2, a1, sb2, st2, sf2, prec, stream));
  fnssl_btf_view v1 = {a1, sb2, st2, sf2};
  FNSSL_SN_TRY(fnssl_sn_full(&v1, nb, nt, nf / 2, &L->full, 1, a1, sb2, st2, sf2, prec, stream));
  strides(nfc, sb, st, sf, nt);
  FNSSL_SN_TRY(fnssl_sn_fconv(&v1, nb, nt, nf / 2, &L->fconv2, 1, 8, a2, sb, st, sf, prec, stream));
  fnssl_btf_view v2 = {a2, sb, st, sf};
  FNSSL_SN_TRY(fnssl_sn_mamba(&v2, nb, nt, nfc, &L->mamba[0], 1, 1, conv_st(0, 0), ssm_st(0, 0), carry, a2, sb, st, sf,
                              mws, mws_bytes, prec, stream));
  if (nt2 == 0) return FNSSL_OK;                 // fewer frames than one pooled step: empty output (AvgPool floor)
  strides(nfc, sb2, st2, sf2, nt2);
  FNSSL_SN_TRY(fnssl_sn_mamba(&v2, nb, nt, nfc, &L->mamba[1], 1, ratio, conv_st(0, 1), ssm_st(0, 1), carry, a3, sb2, st2,
                              sf2, mws, mws_bytes, prec, stream));
  fnssl_btf_view v3 = {a3, sb2, st2, sf2};
  for (int l = 1; l < net->num_layers; ++l) {
    L = &net->layers[l];
    FNSSL_SN_TRY(fnssl_sn_fconv(&v3, nb, nt2, nfc, &L->fconv1, 1, 1, a3, sb2, st2, sf2, prec, stream));
    FNSSL_SN_TRY(fnssl_sn_full(&v3, nb, nt2, nfc, &L->full, 1, a3, sb2, st2, sf2, prec, stream));
    FNSSL_SN_TRY(fnssl_sn_fconv(&v3, nb, nt2, nfc, &L->fconv2, 1, 1, a3, sb2, st2, sf2, prec, stream));
    for (int j = 0; j < 2; ++j)
      FNSSL_SN_TRY(fnssl_sn_mamba(&v3, nb, nt2, nfc, &L->mamba[j], 1, 1, conv_st(l, j), ssm_st(l, j), carry, a3, sb2, st2,
                                  sf2, mws, mws_bytes, prec, stream));
  }
#undef FNSSL_SN_TRY
  return fnssl_sn_head(&v3, nb, nt2, nfc, net->wfiP, net->bfiP, net->wdT, net->bd, out, stream);
}

}  // extern "C"
