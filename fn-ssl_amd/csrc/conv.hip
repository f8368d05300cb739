// Causal 3x3 Conv2d over (frequency, time) as an fp32-MFMA implicit GEMM, plus the time
// average-pool — the CausCnnBlock head of IPDnet (reference IPDnet/FixedAarryIPDnet.py:42-73:
// Conv2d(k=3x3, pad (1,2), bias=False) -> [ReLU] -> crop the last 2 time steps -> AvgPool2d((1,k))).
// Padding 2 in time followed by the crop makes the conv causal: out[t] sees t-2..t.
//
// Same machinery as the LSTM kernel (lstm_kernel.h): activations are channels-last, one wave
// owns 16 consecutive time positions of one (utterance, bin) row and computes
//     out^T [Cout x 16] = W [Cout x 9*Cin] * patches^T [9*Cin x 16]
// with v_mfma_f32_16x16x4_f32 (A = 16-output-channel weight tile, B = activations, lane <->
// position), the packed weight stream (identical for every tile) is shared by the waves of a
// workgroup through the 2-slot LDS ring, A operands are software-pipelined half a quad ahead.
// The input may be the concatenation of two channels-last tensors (the concat skip of the FN
// block is never materialised).
#include <cstring>

#include "lstm_kernel.h"

using namespace fnssl_lstm;

namespace {

struct ConvParams {
  const float* xa;        // segment A [.., CA] channels-last
  const float* xb;        // segment B (may be null)
  long long a_sb, a_sf, a_st, b_sb, b_sf, b_st;   // strides (floats) of batch / bin / time
  int ca, cb;             // channels per segment (ca % 16 == 0; cb % 4 == 0)
  const float* wpack;
  float* out;             // [nb, nf, nt, cout_stride] channels-last, contiguous
  int cout, cout_stride;
  int nb, nf, nt;
  int act;                // 0 none, 1 relu, 2 tanh
  int ntiles, tiles_t;    // tiles = nb * nf * tiles_t, tiles_t = ceil(nt / 16)
  int passes;             // per workgroup
  int quads_per_pass, chq, pad;
  int a_bf16;             // bf16 kernels: segment A holds bf16 (xa addresses 2-byte elements, a_s* count elements)
};

// Stream per pass: for tap (df-major, dt-minor): segment A blocks, segment B blocks (16-channel
// blocks, then 4-channel remainder blocks).  One k-step = NR records (NR = NT/4; one record =
// float4 per lane = the A operands of 4 output-channel tiles); records are grouped in quads of 4,
// a remainder k-step occupies a whole quad.
template <int NT, int NW, int M>
__global__ void __launch_bounds__(NW * 64) conv3x3_kernel(const ConvParams p) {
  static_assert(NT == 4 || NT == 8, "4 or 8 output-channel tiles");
  constexpr int NR = NT / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

  WStream<NW, M, 1> ws;
  ws.nobar = false;
  ws.init(p.wpack, lane, w, p.quads_per_pass, 1, p.chq, p.pad, smem);
  v4f a0 = ws.record(0), a1 = ws.record(1);

  const int nva = p.ca >> 4, nvb = p.cb >> 4, nsb = (p.cb & 15) >> 2;

  for (int pass = 0; pass < p.passes; ++pass) {
    int tile = (pass * (int)gridDim.x + (int)blockIdx.x) * NW + w;
    const bool tvalid = tile < p.ntiles;
    if (!tvalid) tile = p.ntiles - 1;
    const int tt = tile % p.tiles_t;
    const int bf = tile / p.tiles_t;
    const int f = bf % p.nf;
    const int b = __builtin_amdgcn_readfirstlane(bf / p.nf);
    const int t0 = tt * 16;
    // per-utterance descriptors: offsets inside one utterance fit 32 bits
    const rsrc_t ra = make_rsrc(p.xa + (long long)b * p.a_sb);
    const rsrc_t rb = make_rsrc(p.xb ? p.xb + (long long)b * p.b_sb : p.xa);
    const rsrc_t ro = make_rsrc(p.out + (long long)b * p.nf * p.nt * p.cout_stride);
    int tpos = t0 + n;                       // this lane's output position
    const bool pvalid = tvalid && tpos < p.nt;

    v4f acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = v4f{0.f, 0.f, 0.f, 0.f};

    for (int df = -1; df <= 1; ++df) {
      const int ff = f + df;
      const bool frow = ff >= 0 && ff < p.nf;          // wave-uniform: zero padding in frequency
      const int fc = frow ? ff : f;
      for (int dt = 0; dt < 3; ++dt) {
        int ts = tpos + dt - 2;                          // source time of this lane
        const bool live = frow && ts >= 0;               // zero padding on the causal side
        ts = ts < 0 ? 0 : (ts >= p.nt ? p.nt - 1 : ts);
        const unsigned offa = (unsigned)(((long long)fc * p.a_sf + (long long)ts * p.a_st) * 4) + 16 * g;
        const unsigned offb = (unsigned)(((long long)fc * p.b_sf + (long long)ts * p.b_st) * 4);
        // ---- 16-channel blocks of both segments: 4 k-steps each -----------------
        const int nv = nva + nvb;
        v4f xn = v4f{0.f, 0.f, 0.f, 0.f};
        if (nv > 0) xn = nva > 0 ? bld4(ra, offa, 0) : bld4(rb, offb + 16 * g, 0);
        for (int v = 0; v < nv; ++v) {
          v4f xc = xn;
          if (!live) xc = v4f{0.f, 0.f, 0.f, 0.f};
          if (v + 1 < nv) xn = (v + 1 < nva) ? bld4(ra, offa, 64 * (v + 1)) : bld4(rb, offb + 16 * g, 64 * (v + 1 - nva));
          if (NR == 2) {
            // quad = records {k0: tiles 0-3, k0: tiles 4-7, k1: tiles 0-3, k1: tiles 4-7}
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
              const float bk0 = hq == 0 ? xc.x : xc.z, bk1 = hq == 0 ? xc.y : xc.w;
              const v4f a2 = ws.record(2), a3 = ws.record(3);
              __builtin_amdgcn_sched_barrier(0);
              MFMA4(acc, a0, bk0);
              MFMA4((acc + 4), a1, bk0);
              ws.peek_next(a0, a1);
              __builtin_amdgcn_sched_barrier(0);
              MFMA4(acc, a2, bk1);
              MFMA4((acc + 4), a3, bk1);
              if (ws.advance()) {
                a0 = ws.record(0);
                a1 = ws.record(1);
              }
            }
          } else {
            const v4f a2 = ws.record(2), a3 = ws.record(3);
            __builtin_amdgcn_sched_barrier(0);
            MFMA4(acc, a0, xc.x);
            MFMA4(acc, a1, xc.y);
            ws.peek_next(a0, a1);
            __builtin_amdgcn_sched_barrier(0);
            MFMA4(acc, a2, xc.z);
            MFMA4(acc, a3, xc.w);
            if (ws.advance()) {
              a0 = ws.record(0);
              a1 = ws.record(1);
            }
          }
        }
        // ---- 4-channel remainder blocks of segment B: one k-step each ------------
        for (int u = 0; u < nsb; ++u) {
          float xs = bld1(rb, offb + 4 * g, 64 * nvb + 16 * u);
          if (!live) xs = 0.f;
          MFMA4(acc, a0, xs);
          if (NR == 2) MFMA4((acc + 4), a1, xs);
          ws.peek_next(a0, a1);
          if (ws.advance()) {
            a0 = ws.record(0);
            a1 = ws.record(1);
          }
        }
      }
    }
    for (int u = 0; u < p.pad; ++u) {   // ring padding: chunk ends == pass ends
      ws.peek_next(a0, a1);
      if (ws.advance()) {
        a0 = ws.record(0);
        a1 = ws.record(1);
      }
    }
    // ---- epilogue: activation + channels-last store --------------------------------
    if (tpos >= p.nt) tpos = p.nt - 1;
    const unsigned obase = (unsigned)((((long long)f * p.nt + tpos) * p.cout_stride) * 4) + 16 * g;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      v4f r = acc[j];
      if (p.act == 1) {
        r = v4f{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
      } else if (p.act == 2) {
        r = v4f{tanhf(r.x), tanhf(r.y), tanhf(r.z), tanhf(r.w)};
      }
      if (pvalid && 16 * j + 4 * g < p.cout) bst4(r, ro, obase, 64 * j);
    }
  }
}

// bf16-MFMA twin (BASELINE config 3): weights bf16 in the stream, the activation operands rounded to bf16 when
// they enter the MFMA, fp32 accumulation and tensors.  One v_mfma_f32_16x16x32_bf16 per (output tile, PAIR of
// 16-channel blocks); a record = the A operand of one tile for one pair (lane (i, g): output channel 16j + i,
// channels 16*b0 + 4g + {0..3} and 16*b1 + 4g + {0..3}), NT records per pair = NT/4 quads.  Blocks run over the
// concatenation [xa | xb] (both multiples of 16 channels); an odd last block is paired with zeros.
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));

typedef unsigned int v2uc __attribute__((ext_vector_type(2)));

// ABF: segment A arrives as bf16 (the wide bf16 LSTM kernels write bf16 activations): its blocks are ready-made
// operand halves, no conversion and half the bytes.
// PT: 16-position tiles per wave.  With PT = 1 every 1 KiB weight record read from LDS feeds ONE 16-cycle MFMA, i.e.
// 12 waves ask the LDS for 3x its 256 B/clk (round 1: 15 % of the bf16 roof, LDS array 18 % "busy" but the waves parked
// 75 %); with PT = 2 a record feeds two MFMAs (the wave owns 32 consecutive time positions).
template <int NT, int NW, int M, bool ABF = false, int PT = 2>
__global__ void __launch_bounds__(NW * 64) conv3x3_bf16_kernel(const ConvParams p) {
  static_assert(NT == 4 || NT == 8, "4 or 8 output-channel tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

  WStream<NW, M, 1> ws;
  ws.nobar = false;
  ws.init(p.wpack, lane, w, p.quads_per_pass, 1, p.chq, p.pad, smem);
  const int nva = p.ca >> 4, nv = nva + (p.cb >> 4), npairs = (nv + 1) >> 1;
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  const v4bf zb = v4bf{0, 0, 0, 0};

  for (int pass = 0; pass < p.passes; ++pass) {
    int tile = (pass * (int)gridDim.x + (int)blockIdx.x) * NW + w;
    const bool tvalid = tile < p.ntiles;
    if (!tvalid) tile = p.ntiles - 1;
    const int tt = tile % p.tiles_t;
    const int bf = tile / p.tiles_t;
    const int f = bf % p.nf;
    const int b = __builtin_amdgcn_readfirstlane(bf / p.nf);
    const rsrc_t ra = ABF ? make_rsrc(reinterpret_cast<const unsigned short*>(p.xa) + (long long)b * p.a_sb)
                          : make_rsrc(p.xa + (long long)b * p.a_sb);
    const rsrc_t rb = make_rsrc(p.xb ? p.xb + (long long)b * p.b_sb : p.xa);
    const rsrc_t ro = make_rsrc(p.out + (long long)b * p.nf * p.nt * p.cout_stride);
    const int tpos0 = tt * (16 * PT) + n;
    v4f acc[NT][PT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < PT; ++q) acc[j][q] = zero4;

    for (int df = -1; df <= 1; ++df) {
      const int ff = f + df;
      const bool frow = ff >= 0 && ff < p.nf;
      const int fc = frow ? ff : f;
      for (int dt = 0; dt < 3; ++dt) {
        unsigned offa[PT], offb[PT];
        bool live[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) {
          int ts = tpos0 + 16 * q + dt - 2;
          live[q] = frow && ts >= 0;
          ts = ts < 0 ? 0 : (ts >= p.nt ? p.nt - 1 : ts);
          offa[q] = (unsigned)(((long long)fc * p.a_sf + (long long)ts * p.a_st) * (ABF ? 2 : 4)) + (ABF ? 8 : 16) * g;
          offb[q] = (unsigned)(((long long)fc * p.b_sf + (long long)ts * p.b_st) * 4) + 16 * g;
        }
        // 16-channel block v of [xa | xb] as the lane's 4 bf16 operand values (channels 16 v + 4 g + 0..3)
        auto block = [&](int v, int q) -> v4bf {
          if (v >= nv) return zb;
          if (v < nva) {
            if constexpr (ABF)
              return __builtin_bit_cast(v4bf, __builtin_amdgcn_raw_buffer_load_b64(ra, offa[q], 32 * v, 0));
            else
              return __builtin_convertvector(bld4(ra, offa[q], 64 * v), v4bf);
          }
          return __builtin_convertvector(bld4(rb, offb[q], 64 * (v - nva)), v4bf);
        };
        v4bf n0[PT], n1[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) {
          n0[q] = block(0, q);
          n1[q] = block(1, q);
        }
        for (int pi = 0; pi < npairs; ++pi) {
          v8bf bv[PT];
#pragma unroll
          for (int q = 0; q < PT; ++q) {
            const v4bf lo = live[q] ? n0[q] : zb, hi = live[q] ? n1[q] : zb;
            bv[q] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            n0[q] = block(2 * pi + 2, q);   // operands of the next pair: in flight behind this pair's MFMAs
            n1[q] = block(2 * pi + 3, q);
          }
#pragma unroll
          for (int qd = 0; qd < NT / 4; ++qd) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const v8bf av = __builtin_bit_cast(v8bf, ws.record(j));
#pragma unroll
              for (int q = 0; q < PT; ++q)
                acc[4 * qd + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[q], acc[4 * qd + j][q], 0, 0, 0);
            }
            (void)ws.advance();
          }
        }
      }
    }
    for (int u = 0; u < p.pad; ++u) (void)ws.advance();
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      int tpos = tpos0 + 16 * q;
      const bool pvalid = tvalid && tpos < p.nt;
      if (tpos >= p.nt) tpos = p.nt - 1;
      const unsigned obase = (unsigned)((((long long)f * p.nt + tpos) * p.cout_stride) * 4) + 16 * g;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        v4f r = acc[j][q];
        if (p.act == 1) {
          r = v4f{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
        } else if (p.act == 2) {
          r = v4f{tanhf(r.x), tanhf(r.y), tanhf(r.z), tanhf(r.w)};
        }
        if (pvalid && 16 * j + 4 * g < p.cout) bst4(r, ro, obase, 64 * j);
      }
    }
  }
}

// out[b, f, t2, c] = mean_k in[b, f, K*t2 + k, c]   (AvgPool2d((1, K)), floors)
__global__ void __launch_bounds__(256)
pool_t_kernel(const float4* __restrict__ in, int rows, int nt, int c4, int K, float4* __restrict__ out) {
  const int nt2 = nt / K;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over rows * nt2 * c4
  const long long total = (long long)rows * nt2 * c4;
  if (idx >= total) return;
  const int c = (int)(idx % c4);
  const long long rt = idx / c4;
  const int t2 = (int)(rt % nt2);
  const long long row = rt / nt2;
  const float4* src = in + (row * nt + (long long)t2 * K) * c4 + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const float4 v = src[(long long)k * c4];
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  const float d = (float)K;
  out[idx] = make_float4(__fdiv_rn(s.x, d), __fdiv_rn(s.y, d), __fdiv_rn(s.z, d), __fdiv_rn(s.w, d));
}

int conv_nt_tiles(int cout) { return cout <= 64 ? 4 : 8; }

template <int NT>
int quads_per_pass_host(int ca, int cb) {
  constexpr int NR = NT / 4;
  const int nvec = (ca >> 4) + (cb >> 4), nscal = (cb & 15) >> 2;
  // a 16-channel block = 4 k-steps = NR quads; a remainder block = 1 k-step = 1 quad
  return 9 * (nvec * NR + nscal);
}

}  // namespace

extern "C" {

size_t fnssl_conv3x3_packed_floats(int cout, int ca, int cb) {
  if (cout <= 0 || cout > 128 || ca < 0 || cb < 0 || (ca & 15) || (cb & 3) || ca + cb == 0) return 0;
  const int q = conv_nt_tiles(cout) == 8 ? quads_per_pass_host<8>(ca, cb) : quads_per_pass_host<4>(ca, cb);
  return (size_t)q * 4 * 256;
}

int fnssl_conv3x3_pack(const float* w, int cout, int ca, int cb, float* packed) {
  FNSSL_REQUIRE(w && packed, "conv3x3_pack: null pointer");
  const size_t total = fnssl_conv3x3_packed_floats(cout, ca, cb);
  FNSSL_REQUIRE(total > 0, "conv3x3_pack: unsupported sizes (cout %d <= 128, ca %d %% 16, cb %d %% 4)", cout, ca, cb);
  const int cin = ca + cb;
  const int NT = conv_nt_tiles(cout), NR = NT / 4;
  std::memset(packed, 0, total * sizeof(float));
  auto W = [&](int oc, int ci, int df, int dt) -> float {
    return oc < cout ? w[(((size_t)oc * cin + ci) * 3 + df) * 3 + dt] : 0.f;
  };
  float* rec = packed;
  auto emit_kstep = [&](int df, int dt, int cbase, int stride_g, int jj) {
    // one k-step: lane (i, g) holds channel cbase + stride_g * g + jj for output channel 16*tile + i
    for (int r = 0; r < NR; ++r, rec += 256)
      for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q)
          rec[l * 4 + q] = W(16 * (4 * r + q) + (l & 15), cbase + stride_g * (l >> 4) + jj, df, dt);
  };
  for (int df = 0; df < 3; ++df)
    for (int dt = 0; dt < 3; ++dt) {
      const int nv = (ca >> 4) + (cb >> 4);
      for (int v = 0; v < nv; ++v)
        for (int jj = 0; jj < 4; ++jj) emit_kstep(df, dt, 16 * v, 4, jj);     // channel 16v + 4g + jj
      const int nscal = (cb & 15) >> 2;
      for (int u = 0; u < nscal; ++u) {
        float* start = rec;
        emit_kstep(df, dt, ca + 16 * (cb >> 4) + 4 * u, 1, 0);                // channel base + g
        rec = start + 4 * 256;                                                // rest of the quad is padding
      }
    }
  if ((size_t)(rec - packed) != total) {
    fnssl::set_error("conv3x3_pack: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

static unsigned short conv_to_bf16(float f) {   // round to nearest even
  unsigned u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static int bf16_quads_per_pass(int cout, int ca, int cb) {
  const int npairs = ((ca >> 4) + (cb >> 4) + 1) / 2;
  return 9 * npairs * (conv_nt_tiles(cout) / 4);
}

size_t fnssl_conv3x3_packed_floats_bf16(int cout, int ca, int cb) {
  if (cout <= 0 || cout > 128 || ca <= 0 || cb < 0 || (ca & 15) || (cb & 15)) return 0;
  return (size_t)bf16_quads_per_pass(cout, ca, cb) * 4 * 256;
}

int fnssl_conv3x3_pack_bf16(const float* w, int cout, int ca, int cb, float* packed) {
  FNSSL_REQUIRE(w && packed, "conv3x3_pack_bf16: null pointer");
  const size_t total = fnssl_conv3x3_packed_floats_bf16(cout, ca, cb);
  FNSSL_REQUIRE(total > 0, "conv3x3_pack_bf16: unsupported sizes (cout %d <= 128, ca %d and cb %d multiples of 16)", cout,
                ca, cb);
  std::memset(packed, 0, total * sizeof(float));
  const int cin = ca + cb, nv = cin >> 4, NT = conv_nt_tiles(cout);
  float* recf = packed;
  for (int df = 0; df < 3; ++df)
    for (int dt = 0; dt < 3; ++dt)
      for (int pi = 0; pi < (nv + 1) / 2; ++pi)
        for (int j = 0; j < NT; ++j, recf += 256) {   // record = output tile j
          unsigned short* rec = reinterpret_cast<unsigned short*>(recf);
          for (int l = 0; l < 64; ++l)
            for (int half = 0; half < 2; ++half)
              for (int jj = 0; jj < 4; ++jj) {
                const int blk = 2 * pi + half, oc = 16 * j + (l & 15), ci = 16 * blk + 4 * (l >> 4) + jj;
                const float v = (blk < nv && oc < cout) ? w[(((size_t)oc * cin + ci) * 3 + df) * 3 + dt] : 0.f;
                rec[l * 8 + half * 4 + jj] = conv_to_bf16(v);
              }
        }
  if ((size_t)(recf - packed) != total) {
    fnssl::set_error("conv3x3_pack_bf16: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

static int conv_run(int bf, const float* xa, long long a_sb, long long a_sf, long long a_st, int ca, const float* xb,
                    long long b_sb, long long b_sf, long long b_st, int cb, const float* wpack, int cout, int nb,
                    int nf, int nt, int act, float* out, int cout_stride, void* stream) {
  FNSSL_REQUIRE(xa && wpack && out, "conv3x3: null pointer");
  FNSSL_REQUIRE(nb > 0 && nf > 0 && nt > 0, "conv3x3: empty problem");
  FNSSL_REQUIRE((bf ? fnssl_conv3x3_packed_floats_bf16(cout, ca, cb) : fnssl_conv3x3_packed_floats(cout, ca, cb)) > 0 &&
                    ca > 0,
                "conv3x3: unsupported channel counts");
  FNSSL_REQUIRE(cb == 0 || xb, "conv3x3: segment B missing");
  FNSSL_REQUIRE(cout_stride >= cout && cout_stride % 4 == 0 && act >= 0 && act <= 2, "conv3x3: bad output spec");
  auto fits = [&](long long sb, long long sf, long long st, int c) {
    return sb >= 0 && sf >= 0 && st >= 0 && ((long double)nf * sf + (long double)nt * st + c) * 4 < 4.0e9L &&
           !(sb & 3) && !(sf & 3) && !(st & 3);
  };
  FNSSL_REQUIRE(bf != 2 || (reinterpret_cast<uintptr_t>(xa) % 8 == 0), "conv3x3: bf16 segment A must be 8-byte aligned");
  FNSSL_REQUIRE(fits(a_sb, a_sf, a_st, ca) && (cb == 0 || fits(b_sb, b_sf, b_st, cb)) &&
                    (long double)nf * nt * cout_stride * 4 < 4.0e9L,
                "conv3x3: one utterance must be addressable with 32-bit byte offsets; strides multiples of 4 floats");
  ConvParams p;
  p.xa = xa;
  p.xb = cb ? xb : nullptr;
  p.a_sb = a_sb;
  p.a_sf = a_sf;
  p.a_st = a_st;
  p.b_sb = b_sb;
  p.b_sf = b_sf;
  p.b_st = b_st;
  p.ca = ca;
  p.cb = cb;
  p.wpack = wpack;
  p.out = out;
  p.cout = cout;
  p.cout_stride = cout_stride;
  p.nb = nb;
  p.nf = nf;
  p.nt = nt;
  p.act = act;
  p.a_bf16 = bf == 2;
  p.tiles_t = bf ? (nt + 31) / 32 : (nt + 15) / 16;   // bf16 kernels: 32 positions per wave
  p.ntiles = nb * nf * p.tiles_t;
  const int NT = conv_nt_tiles(cout);
  constexpr int NW = 12, M = 4;
  p.quads_per_pass = bf ? bf16_quads_per_pass(cout, ca, cb)
                        : (NT == 8 ? quads_per_pass_host<8>(ca, cb) : quads_per_pass_host<4>(ca, cb));
  Variant vr{NW, M, 1};
  choose_chunk(p.quads_per_pass, vr, p.chq, p.pad);
  const int ncu = fnssl::device_cus();
  const int groups = (p.ntiles + NW - 1) / NW;
  const int nwg = groups < ncu ? groups : ncu;
  p.passes = (groups + nwg - 1) / nwg;
  const size_t lds = (size_t)2 * p.chq * 4096;
  const double flops = 2.0 * 9 * (ca + cb) * (double)cout * nb * nf * (double)nt;
  fnssl::TimedLaunch tl(bf ? "conv3x3_bf16" : "conv3x3", fnssl::as_stream(stream), flops);
  if (bf == 2) {
    if (NT == 8) {
      auto k = conv3x3_bf16_kernel<8, NW, M, true>;
      if (lds > 48 * 1024)
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, fnssl::as_stream(stream), p);
    } else {
      auto k = conv3x3_bf16_kernel<4, NW, M, true>;
      if (lds > 48 * 1024)
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, fnssl::as_stream(stream), p);
    }
  } else if (bf) {
    if (NT == 8) {
      auto k = conv3x3_bf16_kernel<8, NW, M>;
      if (lds > 48 * 1024)
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, fnssl::as_stream(stream), p);
    } else {
      auto k = conv3x3_bf16_kernel<4, NW, M>;
      if (lds > 48 * 1024)
        FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, fnssl::as_stream(stream), p);
    }
  } else if (NT == 8) {
    auto k = conv3x3_kernel<8, NW, M>;
    if (lds > 48 * 1024)
      FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, fnssl::as_stream(stream), p);
  } else {
    auto k = conv3x3_kernel<4, NW, M>;
    if (lds > 48 * 1024)
      FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, fnssl::as_stream(stream), p);
  }
  FNSSL_CHECK_LAUNCH("conv3x3_kernel");
  return FNSSL_OK;
}

int fnssl_conv3x3_causal(const float* xa, long long a_sb, long long a_sf, long long a_st, int ca, const float* xb,
                         long long b_sb, long long b_sf, long long b_st, int cb, const float* wpack, int cout,
                         int nb, int nf, int nt, int act, float* out, int cout_stride, void* stream) {
  return conv_run(0, xa, a_sb, a_sf, a_st, ca, xb, b_sb, b_sf, b_st, cb, wpack, cout, nb, nf, nt, act, out,
                  cout_stride, stream);
}

int fnssl_conv3x3_causal_bf16(const float* xa, long long a_sb, long long a_sf, long long a_st, int ca, const float* xb,
                              long long b_sb, long long b_sf, long long b_st, int cb, const float* wpack, int cout,
                              int nb, int nf, int nt, int act, float* out, int cout_stride, void* stream) {
  return conv_run(1, xa, a_sb, a_sf, a_st, ca, xb, b_sb, b_sf, b_st, cb, wpack, cout, nb, nf, nt, act, out,
                  cout_stride, stream);
}

int fnssl_conv3x3_causal_bf16a(const void* xa_bf16, long long a_sb, long long a_sf, long long a_st, int ca,
                               const float* xb, long long b_sb, long long b_sf, long long b_st, int cb,
                               const float* wpack, int cout, int nb, int nf, int nt, int act, float* out,
                               int cout_stride, void* stream) {
  return conv_run(2, static_cast<const float*>(xa_bf16), a_sb, a_sf, a_st, ca, xb, b_sb, b_sf, b_st, cb, wpack, cout, nb, nf,
                  nt, act, out, cout_stride, stream);
}

int fnssl_avgpool_time(const float* x, int rows, int nt, int c, int k, float* y, void* stream) {
  FNSSL_REQUIRE(x && y && rows > 0 && nt > 0 && c > 0 && c % 4 == 0 && k > 0, "avgpool_time: bad arguments");
  const long long total = (long long)rows * (nt / k) * (c / 4);
  if (total == 0) return FNSSL_OK;
  FNSSL_REQUIRE((total + 255) / 256 < (1ll << 31), "avgpool_time: too large");
  fnssl::TimedLaunch tl("avgpool_time", fnssl::as_stream(stream));
  hipLaunchKernelGGL(pool_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, fnssl::as_stream(stream),
                     reinterpret_cast<const float4*>(x), rows, nt, c / 4, k, reinterpret_cast<float4*>(y));
  FNSSL_CHECK_LAUNCH("pool_t_kernel");
  return FNSSL_OK;
}

}  // extern "C"
