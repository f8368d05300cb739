// Causal 3x3 Conv2d of CausCnnBlock (reference IPDnet/FixedAarryIPDnet.py:42-73) for the bf16 path of BASELINE
// config 3, second formulation: activations are staged through LDS in coalesced half-lines instead of being
// fetched in MFMA-fragment shape.
//
// Why a second kernel: conv3x3_bf16_kernel (conv.hip) loads its B operands straight into registers — lane =
// (position, 4-channel slice), 8 bytes per lane, 32 bytes per position per instruction, each instruction touching
// 16 cache lines and every activation fetched 9 times (once per tap).  It measures 380 TFLOP/s (15 % of the bf16 MFMA
// roof) with the waves parked on vmcnt/barriers; deeper prefetch did not help (the texture-address path, not
// latency, is what saturates).  Here:
//   * a workgroup = 8 waves = 4 position groups (64 consecutive frames of one (utterance, bin) row each) x 2 halves of
//     the 128 output channels; D [32 cout x 32 pos] tiles of v_mfma_f32_32x32x16_bf16, 2 x 2 tiles per wave;
//   * input channels are walked in groups of 64 bytes per position (32 bf16 channels of segment A, or 16 fp32
//     channels of the skip segment B): the 66 x 64 B image of a group is copied once by LDS-DMA (4 lanes per
//     position, 16 positions per instruction -> whole half-lines) into an XOR-swizzled LDS tile and then serves
//     all three time taps and both waves of the pair — 6 k-steps per fetch instead of one;
//   * the packed weight stream (identical for every tile) runs through a 6-slot LDS ring of 8 KiB pieces (one
//     piece = one (df, group, dt) = 2 k-steps x 4 cout tiles), one DMA instruction per wave per piece, 4 pieces
//     in flight, counted s_waitcnt vmcnt(N) + raw s_barrier every second piece;
//   * persistent workgroups; an XCD gets a contiguous range of the tile list so that the bin rows f-1, f, f+1 that
//     share input rows meet in the same L2.
#include <cstdlib>
#include <cstring>

#include "lstm_kernel.h"

using namespace fnssl_lstm;

namespace {

typedef __bf16 v8bfx __attribute__((ext_vector_type(8)));
typedef float v16fx __attribute__((ext_vector_type(16)));

constexpr int kPiece = 8192;     // bytes of one weight piece: 8 records of 1 KiB = one (df, group, dt): 2 k-steps x 4 cout tiles
constexpr int kGroupW = 3 * kPiece;   // weight bytes of one channel group of one bin tap (3 time taps)
constexpr int kXTile = 5120;     // bytes of one activation tile: 72 positions x 64 B (+ pad to 5 DMA instructions)
constexpr int kRW = 2;           // weights are requested this many channel groups ahead
constexpr int kNSLOT = kRW + 1;  // ring slots (of one group each)
constexpr int kXA = 3;           // activation tiles are requested this many channel groups ahead
constexpr int kNBUF = kXA + 1;   // activation tile buffers per position group

struct ConvXParams {
  const void* xa;        // bf16 [.., ca]
  const float* xb;       // fp32 [.., cb] or null
  long long a_sb, a_sf, a_st, b_sb, b_sf, b_st;   // element strides
  unsigned a_bytes, b_bytes;                      // bytes from a bin row's first frame to the end of its last (buffer bound: loads beyond it give 0)
  const void* wpack;
  float* out;
  int cout, cout_stride;
  int nb, nf, nt, act;
  int tiles_t, ntw;      // wave tiles = nb * nf * tiles_t; a tile = tstride output frames
  int pool, out_bf16;    // fused AvgPool2d((1, pool)) of the activated output (1 = none, 3, 4); pooled output as bf16
  int tstride, ntp;      // output frames per tile (63 with pool 3: 21 windows; else 64); frames of the (pooled) output
  int nwt, share, per_pass, passes;   // workgroup tiles, per-XCD share, workgroups per XCD, passes
  int ga, gtot;          // channel groups of segment A (ca / 32) and in total (+ cb / 16)
  int abl;               // make ABLATE=1 builds: timing ablations with wrong results (1 no MFMA, 2 no activation DMA, 4 no weight DMA, 8 no barrier, 16 no LDS reads, 32 no stores)
};

__device__ __forceinline__ rsrc_t make_rsrc_n(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__device__ __forceinline__ void xdma16(rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
  // 1 KiB per wave-instruction: lane l's 16 bytes land at lds + 16 l (not counted by the compiler: vmcnt below)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(r),
               "s"(__builtin_amdgcn_readfirstlane(lds)), "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

struct TileCoord {
  int b, f, tt;
  bool valid;
};

template <bool ABL>
__global__ void __launch_bounds__(512) conv3x3_bf16x_kernel(const ConvXParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int n = lane & 31, hb = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pg = w & 3, ch = w >> 2;   // waves w and w + 4 share a SIMD: one fetches weights, the other activations
  const unsigned lds0 = (unsigned)(uintptr_t)smem;   // LDS byte address of the dynamic segment
  const unsigned xt0 = kNSLOT * kGroupW + pg * (kNBUF * kXTile);   // this position group's tile buffers (offset)
  const int G = p.gtot, NG = 3 * G;                  // channel groups per bin tap and per tile

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  auto coord = [&](int pass) -> TileCoord {
    const int within = pass * p.per_pass + slot;
    const int wgt = xcd * p.share + within;
    const int wt = wgt * 4 + pg;
    TileCoord c;
    c.valid = within < p.share && wgt < p.nwt && wt < p.ntw;
    const int wtc = c.valid ? wt : 0;
    c.tt = wtc % p.tiles_t;
    const int bf = wtc / p.tiles_t;
    c.f = bf % p.nf;
    c.b = bf / p.nf;
    return c;
  };

  // ---- loaders ---------------------------------------------------------------------------------------------------
  // The two waves of a position group split the fetching by kind, because a wave's memory operations complete in
  // order: activation tiles come from HBM on first touch and would hold up the L2-resident weight pieces queued
  // behind them.  Per channel group (= per barrier interval) the ch = 0 wave issues 6 DMA instructions (records
  // 2 pg, 2 pg + 1 of the group's 3 weight pieces, kRW groups ahead), the ch = 1 wave 5 (its position group's
  // activation tile, kXA groups ahead).
  const rsrc_t rw = make_rsrc(p.wpack);
  const unsigned wlane = (unsigned)(pg * 2048 + lane * 16);
  int wl_r = 0, wl_slot = 0;              // group (within the tile) and ring slot of the next weight request
  auto issue_w_group = [&]() {
    if (ABL && (p.abl & 4)) return;
    const unsigned so = (unsigned)wl_r * kGroupW, ld = lds0 + (unsigned)wl_slot * kGroupW + (unsigned)pg * 2048u;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      xdma16(rw, wlane, so + dt * kPiece, ld + dt * kPiece);
      xdma16(rw, wlane, so + dt * kPiece + 1024u, ld + dt * kPiece + 1024u);
    }
    wl_r = wl_r + 1 == NG ? 0 : wl_r + 1;
    wl_slot = wl_slot + 1 == kNSLOT ? 0 : wl_slot + 1;
  };
  // Activation tile of (pass, df, g): positions t0 - 2 .. t0 + 69 of bin row f + df - 1, 64 bytes each.  Lane l of
  // instruction j copies chunk c of position 16 j + l / 4 to LDS slot 64 j + l, c = (l & 3) ^ ((l >> 4) & 3) (the
  // XOR swizzle that keeps the ds_read_b128 of 16 consecutive positions off each other's banks).  Frames outside
  // [0, nt) and rows outside [0, nf) read as zero through the descriptor's bound: one descriptor per (tile, df)
  // whose base is the row and whose extent is the row's frames (0 for a row that does not exist); a negative frame
  // wraps to a huge unsigned offset.
  int xl_pass = 0, xl_df = 0, xl_g = 0, xl_buf = 0;
  rsrc_t xl_ra, xl_rb;
  unsigned xl_va = 0, xl_vb = 0;          // per-lane byte offset of instruction 0 within the row
  const int xc = (lane & 3) ^ ((lane >> 4) & 3);
  auto xl_row = [&]() {
    const TileCoord c = coord(xl_pass < p.passes ? xl_pass : 0);
    const int ff = c.f + xl_df - 1;
    const bool ok = c.valid && xl_pass < p.passes && ff >= 0 && ff < p.nf;
    const int fc = ok ? ff : 0;
    xl_ra = make_rsrc_n(reinterpret_cast<const char*>(p.xa) + ((long long)c.b * p.a_sb + (long long)fc * p.a_sf) * 2,
                        ok ? p.a_bytes : 0u);
    xl_rb = make_rsrc_n(p.xb ? reinterpret_cast<const char*>(p.xb) + ((long long)c.b * p.b_sb + (long long)fc * p.b_sf) * 4
                             : reinterpret_cast<const char*>(p.xa),
                        ok ? p.b_bytes : 0u);
    const int t = p.tstride * c.tt - 2 + (lane >> 2);
    xl_va = (unsigned)(t * (int)p.a_st * 2) + 16u * xc;
    xl_vb = (unsigned)(t * (int)p.b_st * 4) + 16u * xc;
  };
  xl_row();
  auto issue_x_group = [&]() {
    if (ABL && (p.abl & 2)) return;
    const unsigned dst = lds0 + xt0 + (unsigned)xl_buf * kXTile;
    if (xl_g < p.ga) {
      const unsigned so = 64u * xl_g, step = 32u * (unsigned)p.a_st;   // 16 positions of bf16 rows
#pragma unroll
      for (int j = 0; j < 5; ++j) xdma16(xl_ra, xl_va + j * step, so, dst + j * 1024u);
    } else {
      const unsigned so = 64u * (xl_g - p.ga), step = 64u * (unsigned)p.b_st;
#pragma unroll
      for (int j = 0; j < 5; ++j) xdma16(xl_rb, xl_vb + j * step, so, dst + j * 1024u);
    }
    xl_buf = xl_buf + 1 == kNBUF ? 0 : xl_buf + 1;
    if (++xl_g == G) {
      xl_g = 0;
      if (++xl_df == 3) {
        xl_df = 0;
        ++xl_pass;
      }
      xl_row();
    }
  };

  // prologue: the first kRW weight groups / kXA activation tiles; drained once
  if (ch == 0) {
    for (int i = 0; i < kRW; ++i) issue_w_group();
  } else {
    for (int i = 0; i < kXA; ++i) issue_x_group();
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  v16fx acc[2][2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cc][q][r] = 0.f;

  // ---- consumer ------------------------------------------------------------------------------------------------
  // B operand of (dt, q, chunk c): LDS tile byte  pos * 64 + ((c ^ ((pos >> 2) & 3)) << 4),  pos = 32 q + n + dt.
  // bq holds the part without c; the chunk bits are XORed in: c = 2 b2 + hb for a bf16 group (k-step b2 = channels
  // 16 b2 + 8 hb ..), c = 2 hb + b2 for a skip-segment group (16 fp32 channels: the lane's 8 floats are chunks
  // 2 hb, 2 hb + 1; one k-step).
  unsigned bq[3][2];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pos = 32 * q + n + dt;
      bq[dt][q] = (unsigned)(pos * 64 + (((pos >> 2) & 3) << 4));
    }
  const unsigned cbf[2] = {(unsigned)(hb << 4), (unsigned)((2 + hb) << 4)};
  const unsigned cf32[2] = {(unsigned)((2 * hb) << 4), (unsigned)((2 * hb + 1) << 4)};
  const unsigned aoff = (unsigned)((2 * ch) * 1024 + lane * 16);

  struct Operands {
    v8bfx a[2][2], b[2][2];
  };
  int c_slot = 0, c_buf = 0, c_g = 0, c_df = 0, c_pass = 0;
  // D tiles are [32 positions x 32 output channels]: lane = (channel n, hb), register r = position
  // (r & 3) + 8 (r >> 2) + 4 hb of the tile, so one dword store instruction writes two whole 128-byte lines (32
  // consecutive channels of two positions) — 16-byte stores scattered over 64 rows made every line a partial write.
  // Frames >= nt fall outside the row descriptor's extent and are dropped; channels >= cout get an offset outside it.
  // The time pooling that follows the convolution in CausCnnBlock is applied here (same operation order and
  // roundings as pool_t_kernel: activation, left-to-right sum, IEEE division, optional bf16 rounding):
  //   pool 4: a window is the 4 registers (r & 3) of one (q, r >> 2) — 8 windows per lane, nothing crosses lanes;
  //   pool 3: a tile is 63 frames = 21 windows; the two half-waves exchange their registers (lane ^ 32), then the
  //           hb = 0 half produces the even windows and the hb = 1 half the odd ones.
  auto store_tile = [&]() {
    const TileCoord cc0 = coord(c_pass);
    const unsigned es = p.out_bf16 ? 2u : 4u;
    const unsigned rowb = (unsigned)p.ntp * (unsigned)p.cout_stride * es;
    const rsrc_t ro = make_rsrc_n(reinterpret_cast<const char*>(p.out) +
                                      ((long long)cc0.b * p.nf + cc0.f) * p.ntp * p.cout_stride * es,
                                  cc0.valid ? rowb : 0u);
    const unsigned cse = (unsigned)p.cout_stride * es;
    auto act = [&](float v) -> float {
      if (p.act == 1) return fmaxf(v, 0.f);
      if (p.act == 2) return tanhf(v);
      return v;
    };
    auto put = [&](float v, unsigned vo, unsigned so) {
      if (p.out_bf16)
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), ro, vo, so, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo, so, 0);
    };
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int co = 32 * (2 * ch + cc) + n;
      const bool cok = co < p.cout;
      if (p.pool == 1) {
        const unsigned vo = cok ? (unsigned)(64 * cc0.tt + 4 * hb) * cse + es * co : 0xffffff00u;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            put(act(acc[cc][q][r]), vo, (unsigned)(32 * q + (r & 3) + 8 * (r >> 2)) * cse);
      } else if (p.pool == 4) {
        const unsigned vo = cok ? (unsigned)(16 * cc0.tt + hb) * cse + es * co : 0xffffff00u;   // window 8 q + 2 m + hb
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            float sum = act(acc[cc][q][4 * m]);
#pragma unroll
            for (int k = 1; k < 4; ++k) sum += act(acc[cc][q][4 * m + k]);
            put(__fdiv_rn(sum, 4.f), vo, (unsigned)(8 * q + 2 * m) * cse);
          }
      } else {
        float own[2][16], oth[2][16];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            own[q][r] = act(acc[cc][q][r]);
            oth[q][r] = __shfl_xor(own[q][r], 32, 64);
          }
        // frame p of the tile seen from a lane of half H: own registers if the frame's half ((p >> 2) & 1) is H
        auto frame = [&](int pp, int H) -> float {
          const int q = pp >> 5, i = pp & 31, r = (i & 3) + 4 * (i >> 3);
          return ((i >> 2) & 1) == H ? own[q][r] : oth[q][r];
        };
        const unsigned vo = cok ? (unsigned)(21 * cc0.tt + hb) * cse + es * co : 0xffffff00u;    // window 2 s + hb
#pragma unroll
        for (int sl = 0; sl < 11; ++sl) {
          const float a = (frame(6 * sl, 0) + frame(6 * sl + 1, 0)) + frame(6 * sl + 2, 0);
          float b = 0.f;
          if (sl < 10) b = (frame(6 * sl + 3, 1) + frame(6 * sl + 4, 1)) + frame(6 * sl + 5, 1);
          // window 21 of the hb = 1 half (sl = 10) would belong to the next tile: parked outside the row
          put(__fdiv_rn(hb ? b : a, 3.f), (sl == 10 && hb) ? 0xffffff00u : vo, (unsigned)(2 * sl) * cse);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cc][q][r] = 0.f;
    }
  };

  // One barrier interval per channel group: the counted wait (a weight wave may have the previous interval's 6
  // instructions outstanding, an activation wave the previous kXA - 1 intervals' 5 each) + barrier make the
  // group's 3 weight pieces and its activation tile readable and the buffers of the previous group free; the
  // interval's DMA instructions go out first, then 3 pieces (time taps) of 8 LDS reads + 8 MFMAs per wave.
  const int total = p.passes * NG;
  for (int gi = 0; gi < total; ++gi) {
    if (gi > 0 && !(ABL && (p.abl & 8))) {
      if (ch == 0)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(6 * (kRW - 1)) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(5 * (kXA - 1)) : "memory");
    }
    if (ch == 0) issue_w_group();

    const bool f32 = c_g >= p.ga;
    const char* ring = smem + c_slot * kGroupW + aoff;
    const char* xt = smem + xt0 + c_buf * kXTile;
    const unsigned cx0 = f32 ? cf32[0] : cbf[0], cx1 = f32 ? cf32[1] : cbf[1];
    auto read_piece = [&](int dt, Operands& o) {
      if (ABL && (p.abl & 16)) {
        asm volatile("" : "+v"(o.a[0][0]), "+v"(o.a[0][1]), "+v"(o.a[1][0]), "+v"(o.a[1][1]), "+v"(o.b[0][0]), "+v"(o.b[0][1]), "+v"(o.b[1][0]), "+v"(o.b[1][1]));
        return;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        o.b[0][q] = *reinterpret_cast<const v8bfx*>(xt + (bq[dt][q] ^ cx0));
        o.b[1][q] = *reinterpret_cast<const v8bfx*>(xt + (bq[dt][q] ^ cx1));
      }
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
          o.a[b2][cc] = *reinterpret_cast<const v8bfx*>(ring + dt * kPiece + (b2 * 4 + cc) * 1024);
    };
    // A skip-segment group: the lane's 8 floats become the b2 = 0 operand; the b2 = 1 MFMAs run on zeros (their
    // stream records are zero as well).  Kept free of control flow around the MFMAs: with the accumulators
    // flowing through a branch the register allocator stops accumulating in place and triples the tile registers.
    auto mfma_piece = [&](Operands& o) {
      if (f32) {
        typedef __bf16 v4bfx __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const v4bfx l4 = __builtin_convertvector(__builtin_bit_cast(v4f, o.b[0][q]), v4bfx);
          const v4bfx h4 = __builtin_convertvector(__builtin_bit_cast(v4f, o.b[1][q]), v4bfx);
          o.b[0][q] = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
          o.b[1][q] = __builtin_bit_cast(v8bfx, v4f{0.f, 0.f, 0.f, 0.f});
        }
      }
      if (ABL && (p.abl & 1)) {
        asm volatile("" ::"v"(o.a[0][0]), "v"(o.a[0][1]), "v"(o.a[1][0]), "v"(o.a[1][1]), "v"(o.b[0][0]), "v"(o.b[0][1]), "v"(o.b[1][0]), "v"(o.b[1][1]));
        return;
      }
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            acc[cc][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.b[b2][q], o.a[b2][cc], acc[cc][q], 0, 0, 0);
    };
    Operands o0, o1;
    read_piece(0, o0);
    read_piece(1, o1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_piece(o0);
    __builtin_amdgcn_sched_barrier(0);
    if (ch != 0) issue_x_group();     // staggered against the SIMD's weight wave, which issues right after the barrier
    read_piece(2, o0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_piece(o1);
    mfma_piece(o0);

    c_slot = c_slot + 1 == kNSLOT ? 0 : c_slot + 1;
    c_buf = c_buf + 1 == kNBUF ? 0 : c_buf + 1;
    if (++c_g == G) {
      c_g = 0;
      if (++c_df == 3) {
        c_df = 0;
        store_tile();
        ++c_pass;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may land after the workgroup has released its LDS
}

// y[row, t2, c] = bf16( mean_k x[row, K t2 + k, c] ): the pooled tensor the next conv reads as its bf16 segment A
__global__ void __launch_bounds__(256)
pool_t_bf16_kernel(const float4* __restrict__ in, int rows, int nt, int c4, int K, uint2* __restrict__ out) {
  const int nt2 = nt / K;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
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
  typedef __bf16 v4bfx __attribute__((ext_vector_type(4)));
  const v4f m = v4f{__fdiv_rn(s.x, d), __fdiv_rn(s.y, d), __fdiv_rn(s.z, d), __fdiv_rn(s.w, d)};
  out[idx] = __builtin_bit_cast(uint2, __builtin_convertvector(m, v4bfx));
}

unsigned short to_bf16_host(float f) {   // round to nearest even
  unsigned u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int groups_of(int ca, int cb) { return ca / 32 + cb / 16; }

}  // namespace

extern "C" {

size_t fnssl_conv3x3_packed_bytes_bf16x(int cout, int ca, int cb) {
  if (cout <= 64 || cout > 128 || (cout & 3) || ca <= 0 || (ca & 31) || cb < 0 || (cb & 15)) return 0;
  return (size_t)9 * groups_of(ca, cb) * kPiece;
}

int fnssl_conv3x3_pack_bf16x(const float* w, int cout, int ca, int cb, void* packed) {
  FNSSL_REQUIRE(w && packed, "conv3x3_pack_bf16x: null pointer");
  const size_t total = fnssl_conv3x3_packed_bytes_bf16x(cout, ca, cb);
  FNSSL_REQUIRE(total > 0,
                "conv3x3_pack_bf16x: unsupported sizes (64 < cout %d <= 128, cout %% 4 == 0, ca %d %% 32 == 0, cb %d %% 16 == 0)",
                cout, ca, cb);
  std::memset(packed, 0, total);
  const int cin = ca + cb, ga = ca / 32, G = groups_of(ca, cb);
  unsigned short* rec = static_cast<unsigned short*>(packed);
  // piece (df, g, dt) = 2 k-steps x 4 cout tiles; record lane l, element i = W[32 ct + l % 32][c0 + 8 (l / 32) + i]
  for (int df = 0; df < 3; ++df)
    for (int g = 0; g < G; ++g)
      for (int dt = 0; dt < 3; ++dt)
        for (int b2 = 0; b2 < 2; ++b2)
          for (int ct = 0; ct < 4; ++ct, rec += 512) {
            int c0;
            if (g < ga) {
              c0 = 32 * g + 16 * b2;
            } else {
              if (b2) continue;             // a skip-segment group has one k-step; its second record set stays 0
              c0 = ca + 16 * (g - ga);
            }
            for (int l = 0; l < 64; ++l)
              for (int i = 0; i < 8; ++i) {
                const int oc = 32 * ct + (l & 31), ci = c0 + 8 * (l >> 5) + i;
                rec[l * 8 + i] = oc < cout ? to_bf16_host(w[(((size_t)oc * cin + ci) * 3 + df) * 3 + dt]) : 0;
              }
          }
  if ((size_t)(reinterpret_cast<char*>(rec) - static_cast<char*>(packed)) != total) {
    fnssl::set_error("conv3x3_pack_bf16x: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

int fnssl_conv3x3_causal_bf16x(const void* xa_bf16, long long a_sb, long long a_sf, long long a_st, int ca,
                               const float* xb, long long b_sb, long long b_sf, long long b_st, int cb,
                               const void* wpack, int cout, int nb, int nf, int nt, int act, int pool, int out_bf16,
                               void* out, int cout_stride, void* stream) {
  FNSSL_REQUIRE(xa_bf16 && wpack && out, "conv3x3_bf16x: null pointer");
  FNSSL_REQUIRE((pool == 1 || pool == 3 || pool == 4) && (out_bf16 == 0 || out_bf16 == 1),
                "conv3x3_bf16x: pool must be 1, 3 or 4 and out_bf16 0 or 1");
  FNSSL_REQUIRE(nb > 0 && nf > 0 && nt > 0, "conv3x3_bf16x: empty problem");
  FNSSL_REQUIRE(fnssl_conv3x3_packed_bytes_bf16x(cout, ca, cb) > 0, "conv3x3_bf16x: unsupported channel counts");
  FNSSL_REQUIRE(cb == 0 || xb, "conv3x3_bf16x: segment B missing");
  FNSSL_REQUIRE(cout_stride >= cout && cout_stride % 4 == 0 && act >= 0 && act <= 2, "conv3x3_bf16x: bad output spec");
  FNSSL_REQUIRE(reinterpret_cast<uintptr_t>(xa_bf16) % 16 == 0 && (cb == 0 || reinterpret_cast<uintptr_t>(xb) % 16 == 0),
                "conv3x3_bf16x: inputs must be 16-byte aligned");
  // extent of one bin row (its nt frames) = the bound of the row's buffer descriptor; the whole utterance must stay
  // below 2 GB for the 32-bit offsets
  auto slab = [&](long long sb, long long sf, long long st, int c, int es, unsigned& bytes) {
    if (sb < 0 || sf < 0 || st < 0 || ((sb * es) & 15) || ((sf * es) & 15) || ((st * es) & 15)) return false;
    if (((long double)(nf - 1) * sf + (long double)(nt - 1) * st + c) * es >= 2.0e9L) return false;
    bytes = (unsigned)(((nt - 1) * st + c) * es);
    return true;
  };
  ConvXParams p;
  FNSSL_REQUIRE(slab(a_sb, a_sf, a_st, ca, 2, p.a_bytes) && (cb == 0 || slab(b_sb, b_sf, b_st, cb, 4, p.b_bytes)) &&
                    (long double)nf * (nt / pool) * cout_stride * 4 < 4.0e9L,
                "conv3x3_bf16x: strides must keep 16-byte alignment and one utterance below 2 GB");
  if (cb == 0) p.b_bytes = 0;
  p.xa = xa_bf16;
  p.xb = cb ? xb : nullptr;
  p.a_sb = a_sb;
  p.a_sf = a_sf;
  p.a_st = a_st;
  p.b_sb = b_sb;
  p.b_sf = b_sf;
  p.b_st = b_st;
  p.wpack = wpack;
  p.out = static_cast<float*>(out);
  p.cout = cout;
  p.cout_stride = cout_stride;
  p.nb = nb;
  p.nf = nf;
  p.nt = nt;
  p.act = act;
  p.pool = pool;
  p.out_bf16 = out_bf16;
  p.ntp = nt / pool;
  if (p.ntp == 0) return FNSSL_OK;                      // fewer frames than one pooling window: empty output
  p.tstride = pool == 3 ? 63 : 64;
  p.tiles_t = (p.ntp * pool + p.tstride - 1) / p.tstride;   // frames past the last whole window are not computed
  const long long ntw = (long long)nb * nf * p.tiles_t;
  FNSSL_REQUIRE(ntw < (1ll << 29), "conv3x3_bf16x: too many tiles");
  p.ntw = (int)ntw;
  p.nwt = (p.ntw + 3) / 4;
  p.share = (p.nwt + 7) / 8;
  const int ncu = fnssl::device_cus();
  int per = ncu / 8 > 0 ? ncu / 8 : 1;
  if (per > p.share) per = p.share;
  p.per_pass = per;
  p.passes = (p.share + per - 1) / per;
  p.ga = ca / 32;
  p.gtot = groups_of(ca, cb);
  const size_t lds = (size_t)kNSLOT * kGroupW + 4 * kNBUF * kXTile;
  const double flops = 2.0 * 9 * (ca + cb) * (double)cout * nb * nf * (double)nt;
  fnssl::TimedLaunch tl("conv3x3_bf16x", fnssl::as_stream(stream), flops);
#ifdef FNSSL_BUILD_ABLATE
  p.abl = getenv("FNSSL_CONVX_ABL") ? atoi(getenv("FNSSL_CONVX_ABL")) : 0;
  auto k = p.abl ? conv3x3_bf16x_kernel<true> : conv3x3_bf16x_kernel<false>;
#else
  p.abl = 0;
  auto k = conv3x3_bf16x_kernel<false>;
#endif
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(8 * per), dim3(512), lds, fnssl::as_stream(stream), p);
  FNSSL_CHECK_LAUNCH("conv3x3_bf16x_kernel");
  return FNSSL_OK;
}

int fnssl_avgpool_time_bf16(const float* x, int rows, int nt, int c, int k, void* y_bf16, void* stream) {
  FNSSL_REQUIRE(x && y_bf16 && rows > 0 && nt > 0 && c > 0 && c % 4 == 0 && k > 0, "avgpool_time_bf16: bad arguments");
  const long long total = (long long)rows * (nt / k) * (c / 4);
  if (total == 0) return FNSSL_OK;
  FNSSL_REQUIRE((total + 255) / 256 < (1ll << 31), "avgpool_time_bf16: too large");
  fnssl::TimedLaunch tl("avgpool_time", fnssl::as_stream(stream));
  hipLaunchKernelGGL(pool_t_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, fnssl::as_stream(stream),
                     reinterpret_cast<const float4*>(x), rows, nt, c / 4, k, reinterpret_cast<uint2*>(y_bf16));
  FNSSL_CHECK_LAUNCH("pool_t_bf16_kernel");
  return FNSSL_OK;
}

}  // extern "C"
