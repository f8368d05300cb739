// "Wide" bf16-MFMA LSTM recurrence: 32 sequences per wave on v_mfma_f32_32x32x16_bf16 (BASELINE config 3).
//
// Why a second bf16 kernel (profiles/r02/c_pmc_ipdnet_bf16.json): lstm_bf16.h reads every weight from
// LDS once per step for only 16 sequences (1 KiB of ds_read per 16-cycle MFMA = the whole 256 B/clk LDS port at
// 4 waves per CU), stages the stream through VGPRs (ds_write_b128 costs 13 LDS cycles per KiB) one chunk ahead,
// and sits at 20 % MFMA-busy with its waves parked 41 % of the time.  Here
//   * a wave owns 32 sequences: one 1 KiB A record (32 gate rows x 16 channels) feeds a 32-cycle MFMA, so LDS reads
//     per flop halve and one wave per SIMD can keep its matrix pipe busy;
//   * the weight stream goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no
//     ds_write) into a 3-slot ring of whole gate-row TILES (consuming T, T + 1 complete, T + 2 in flight: a whole
//     tile-time, > 1000 cycles, of cover for an L2 round trip), one workgroup barrier per tile, completion by a
//     hand-placed s_waitcnt vmcnt(0); the A operands are read from LDS AD = 4 records ahead of their MFMA, across
//     tile boundaries (without that depth every MFMA waits out an LDS round trip: measured 95 k cycles per step
//     against 35 k of MFMA issue);
//   * activations travel between layers as bf16 (they are rounded to bf16 on entry to the next MFMA anyway, so
//     the numbers are those of the fp32-in-HBM path) — x_t arrives as ready-made B operands, no conversion, half
//     the HBM bytes;
//   * the bias rides in the product: a constant-one "block" whose three live columns carry bias = hi + mid + lo
//     (three bf16 terms reproduce the fp32 bias exactly), consumed first, so every accumulator starts at +0;
//   * h_{t-1} (packed bf16 B operands) and the cell state (fp32) live in registers: the D fragment of tile T holds
//     the four gates of units 8T + 4*hb + j for the lane's own sequence, and the K order of the recurrent blocks
//     is permuted in the packer so that a lane's own outputs ARE its B operand of the next step; h_t is parked in
//     a lane-private LDS area during the step (8 bytes per tile) and reloaded as 16 operand registers at its end,
//     which frees the 64 registers a second operand set would take for the A-operand pipeline.
//
// Stream layout (fnssl_lstm_pack_bf16w): tile T (32 rows = [gate 0..3][unit 8T .. 8T+7]) = KT records of 1 KiB,
//   record 0          ones block   (k-slot (hb 0, i 0..2) = bias hi, mid, lo; everything else 0)
//   records 1..NKX    input blocks (16 channels each: src0 blocks, then src2 blocks), k = 16 b + 8 hb + i
//   then NKH records  recurrent blocks s: k-slot (hb, i) <-> unit 16 s + 8 (i >> 2) + 4 hb + (i & 3)
// A-operand element i of lane (row = lane & 31, hb = lane >> 5).
#pragma once

#include "lstm_static.h"

namespace fnssl_lstm {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bfw __attribute__((ext_vector_type(8)));
typedef __bf16 v4bfw __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// element types of the activation tensors of one launch (fnssl_lstm_desc.f32_mask)
constexpr int kW_F0 = 1, kW_F2 = 2, kW_OUTF = 4;   // bit set = that tensor is fp32

__host__ __device__ inline int bf16w_records_per_tile(int c0, int c2, int H) { return 1 + (c0 >> 4) + (c2 >> 4) + (H >> 4); }

__device__ __forceinline__ void dma16(rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
  // one wave-instruction = 1 KiB: lane l's 16 bytes land at lds + 16 l.  M0 is written in the same statement that
  // reads it (the compiler does not preserve it); hipcc does not count this load: see the vmcnt(N) below.
  // (readfirstlane: the operands are wave-uniform by construction; this makes the compiler keep them in SGPRs)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(r),
               "s"(__builtin_amdgcn_readfirstlane(lds)), "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

// split_addr for a tensor of ES-byte elements: wave-uniform base (the wave's minimum) + per-lane byte offset
template <int ES>
__device__ __forceinline__ rsrc_t split_addr_e(const void* base, long long off_elems, unsigned& voff) {
  long long mn = off_elems;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long o = __shfl_xor(mn, d, 64);
    mn = o < mn ? o : mn;
  }
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(mn & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)mn >> 32));
  const long long mnu = (long long)(((unsigned long long)hi << 32) | lo);
  voff = (unsigned)((off_elems - mnu) * ES);
  return make_rsrc(reinterpret_cast<const char*>(base) + mnu * ES);
}

// The same with a compile-time byte offset IMM added to both the stream offset and the LDS destination inside the
// statement (two SALU adds with a literal instead of four instructions of address arithmetic per record).
template <int IMM>
__device__ __forceinline__ void dma16_imm(rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
  unsigned tmp;
  asm volatile("s_add_u32 m0, %3, %5\n\ts_add_u32 %0, %4, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
               : "=&s"(tmp)
               : "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(lds)), "s"(__builtin_amdgcn_readfirstlane(soff)),
                 "n"(IMM)
               : "memory", "scc");   // s_add_u32 writes SCC: without the clobber the compiler keeps a loop condition in it
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  // all but my N youngest vector-memory operations have completed (in order), then the workgroup barrier: every
  // wave's share of the tile that is about to be consumed has landed, and everybody is done with the slot that the
  // next DMA overwrites
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

__device__ __forceinline__ v8bfw join8(v4bfw lo, v4bfw hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// NW consumer waves (32 sequences each) + NL loader waves.  NL = 0: the consumers fetch the stream themselves.  NL > 0
// (launches that leave SIMDs idle anyway — IPDnet's narrow-band layers at batch 64 are 2 groups per CU): the DMA issue
// (4 instructions per KiB) moves to waves on the idle SIMDs and the consumers' issue slots go to MFMAs and gate math.
// ABL (make ABLATE=1 builds only): timing ablations with wrong results — 1 no gate math, 2 no barriers, 4 no MFMAs,
// 8 no A-operand LDS reads, 16 no output stores / h staging, 32 no x loads
template <int H, int NW, int NB0, int NB2, int FLAGS, int NSLOT, int NL, int ABL = 0>
__global__ void __launch_bounds__((NW + NL) * 64) lstm_bf16w_kernel(const LstmParams p) {
  constexpr int NT = H / 8, NKH = H / 16, NKX = NB0 + NB2, KT = 1 + NKX + NKH;
  constexpr int NF = NL > 0 ? NL : NW;                    // waves that fetch
  constexpr int KP = KT / 2;                              // ring granule: half a tile ("piece")
  constexpr int KPW = (KP + NF - 1) / NF;                 // records each fetching wave requests per piece
  constexpr int NPS = 2 * NT;                             // pieces per step
  static_assert(KT % 2 == 0, "tiles are fetched in two pieces");
  constexpr bool F0 = FLAGS & kW_F0, F2 = FLAGS & kW_F2, OUTF = FLAGS & kW_OUTF;
  constexpr int AD = 4;                                   // A-operand reads in flight ahead of the MFMA that uses them
  constexpr int RING = NSLOT * KT * 1024, HBUF = NKH * 1024;   // bytes: weight ring; one wave's h_t staging area
  static_assert(NSLOT == 3, "protocol below: consuming T, T + 1 landed, T + 2 in flight");
  static_assert(RING + NW * HBUF <= 160 * 1024, "ring + h staging do not fit the LDS");
  static_assert(AD < KT, "prefetch depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int n = lane & 31, hb = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int task = p.task0 + wg * NW + w;                 // 32-sequence group
  int q = task * 32 + n;
  const bool valid = q < p.nseq && task < p.task1;
  if (q >= p.nseq) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  // activations: element sizes by template flag; a lane reads 8 consecutive channels (k = 8 hb + i) per block
  constexpr int E0 = F0 ? 4 : 2, E2 = F2 ? 4 : 2, EO = OUTF ? 4 : 2;
  unsigned vo0 = 0, vo2 = 0, voo = 0;
  rsrc_t rx0 = make_rsrc(p.out), rx2 = make_rsrc(p.out);
  if constexpr (NB0 > 0) rx0 = split_addr_e<E0>(p.src0.p, qo * p.src0.so + qi * p.src0.si + 8 * hb, vo0);
  if constexpr (NB2 > 0) rx2 = split_addr_e<E2>(p.src2.p, qo * p.src2.so + qi * p.src2.si + 8 * hb, vo2);
  const rsrc_t ro = split_addr_e<EO>(p.out, qo * p.out_so + qi * p.out_si + dir * H + 4 * hb, voo);
  const unsigned st0 = (unsigned)(p.src0.st * E0), st2 = (unsigned)(p.src2.st * E2), sto = (unsigned)(p.out_st * EO);
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const unsigned lds0 = (unsigned)(size_t)smem;            // LDS byte address of the ring

  // ---- weight ring: NSLOT tiles = 6 half-tile pieces of KP KiB, one barrier per piece.  At the barrier that opens
  // piece j: piece j + 1 has landed everywhere (so the A-operand prefetch may run across the boundary), pieces
  // j + 2 .. j + 4 are in flight (1.5 tile-times, > 1500 cycles, of cover for an L2 round trip under load — with
  // whole tiles and one request in flight per loader the ring, not the MFMAs, set the pace: 2400 cycles per tile),
  // and everybody is done with piece j - 1, whose slot the request for piece j + 5 takes.  A fetching wave certifies
  // its share with s_waitcnt vmcnt(3 * KPW): loads complete in order, so all but its youngest 3 * KPW are done.
  const int fw = NL > 0 ? w - NW : w;                      // my index among the fetching waves (< 0: I do not fetch)
  const int rlast = fw + (KPW - 1) * NF < KP ? fw + (KPW - 1) * NF : KP - 1;   // past the end: repeat the last record
  auto fetch_piece = [&](int piece_in_step, int slot) {   // my share: records fw, fw + NF, ... of the piece
    const unsigned sb = (unsigned)(piece_in_step * KP) * 1024u, lb = lds0 + (unsigned)(slot * KP) * 1024u;
    static_for<KPW - 1>([&](auto mc) {
      dma16_imm<decltype(mc)::value * NF * 1024>(rw, vlane, sb + (unsigned)fw * 1024u, lb + (unsigned)fw * 1024u);
    });
    dma16_imm<0>(rw, vlane, sb + (unsigned)rlast * 1024u, lb + (unsigned)rlast * 1024u);
  };
  int ft = 0, fslot = 0;                                   // next piece to request (position inside the step), its slot
  auto fetch_next = [&]() {
    fetch_piece(ft, fslot);
    ft = ft + 1 == NPS ? 0 : ft + 1;
    fslot = fslot + 1 == 2 * NSLOT ? 0 : fslot + 1;
  };
  auto fetch_barrier = [&]() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * KPW) : "memory"); };
  if (fw >= 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) fetch_next();
  }
  if constexpr (NL > 0) {
    if (w >= NW) {   // ---- loader wave: the same barrier sequence as the consumers, nothing but DMA in between
      const long long npieces = (long long)NPS * p.nsteps;
      for (long long g = 0; g < npieces; ++g) {
        if constexpr (ABL & 2)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * KPW) : "memory");
        else
          fetch_barrier();
        fetch_next();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
  }
  int cslot = 0;                                           // slot (in tiles) of the tile being consumed

  // ---- state ---------------------------------------------------------------------------------------------------
  const v4bfw zb4 = v4bfw{0, 0, 0, 0};
  v8bfw ones;                                              // B operand of the ones block: k-slots (hb 0, i 0..2) = 1
  {
    const __bf16 o1 = (__bf16)(hb == 0 ? 1.0f : 0.0f);
    ones = v8bfw{o1, o1, o1, 0, 0, 0, 0, 0};
  }
  v8bfw xb[NKX > 0 ? NKX : 1], xn[NKX > 0 ? NKX : 1];     // this step's / the next step's input operands
  v8bfw hop[NKH];                                          // h_{t-1} as B operands; h_t is staged in LDS (hbuf)
  v4f creg[NT];
#pragma unroll
  for (int i = 0; i < NKH; ++i) hop[i] = join8(zb4, zb4);
#pragma unroll
  for (int i = 0; i < NT; ++i) creg[i] = v4f{0.f, 0.f, 0.f, 0.f};
  char* const hbuf = smem + RING + w * HBUF + lane * 16;   // [block s][lane][16 B]: tile 2s -> bytes 0-7, 2s+1 -> 8-15

  // input block b of time tt -> B operand (8 channels 16 b + 8 hb + i of the lane's sequence)
  auto load_x = [&](auto bc, unsigned tt) -> v8bfw {
    constexpr int B = decltype(bc)::value;
    if constexpr (B < NB0) {
      if constexpr (F0) {
        const v4f a = bld4(rx0, vo0, tt * st0 + 64 * B), c = bld4(rx0, vo0, tt * st0 + 64 * B + 16);
        return join8(__builtin_convertvector(a, v4bfw), __builtin_convertvector(c, v4bfw));
      } else {
        return __builtin_bit_cast(v8bfw, bld4(rx0, vo0, tt * st0 + 32 * B));
      }
    } else {
      constexpr int B2 = B - NB0;
      if constexpr (F2) {
        const v4f a = bld4(rx2, vo2, tt * st2 + 64 * B2), c = bld4(rx2, vo2, tt * st2 + 64 * B2 + 16);
        return join8(__builtin_convertvector(a, v4bfw), __builtin_convertvector(c, v4bfw));
      } else {
        return __builtin_bit_cast(v8bfw, bld4(rx2, vo2, tt * st2 + 32 * B2));
      }
    }
  };
  static_for<NKX>([&](auto b) { xb[decltype(b)::value] = load_x(b, rev ? p.nsteps - 1 : 0); });

  const char* const lds_rd = smem + lane * 16;
  auto arec = [&](const char* base, int r) { return __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(base + r * 1024)); };
  // With loader waves the consumers never wait on vector memory at a tile boundary (their x loads and h stores are
  // waited for where the compiler needs them); without, a wave also certifies its own DMA share.
  auto piece_barrier = [&]() {
    if constexpr (ABL & 2) return;
    if constexpr (NL > 0) {
      asm volatile("s_barrier" ::: "memory");
    } else {
      fetch_barrier();
      fetch_next();
    }
  };
  // the first piece of all: wait for it, publish, and prime the A pipeline
  piece_barrier();
  v8bfw apipe[AD];
#pragma unroll
  for (int i = 0; i < AD; ++i) apipe[i] = arec(lds_rd, i);

  // Software pipeline over tiles: the gate math of tile T - 1 (VALU: 20 exp, 20 rcp, ~70 others per lane) is issued
  // in the shadow of tile T's MFMAs — its result is only needed by the NEXT step — so the matrix pipe does not idle
  // while a tile's D fragment is post-processed.  accp = the finished accumulator of the previous tile.
  v16f accp = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // NL == 0 only: the h store of a tile is issued one tile later still (right after a barrier), so that the
  // s_waitcnt vmcnt(0) of the following barrier finds it a whole tile-time old
  v4f pend_f = {0.f, 0.f, 0.f, 0.f};
  v4bfw pend_b = zb4;
  unsigned pend_off = 0;
  bool pend_live = false;
  auto gate = [&](auto tp, unsigned oo_t) {     // post-process tile TP (of the step whose output offset is oo_t)
    constexpr int TP = decltype(tp)::value;
    if constexpr (ABL & 1) {
      creg[TP] += v4f{accp[0], accp[5], accp[10], accp[15]};
      return;
    }
    // gates of units 8 TP + 4 hb + j (j = 0..3) of the lane's sequence: D register 4 * gate + j
    const v4f ig = sigmoid4(v4f{accp[0], accp[1], accp[2], accp[3]});
    const v4f fg = sigmoid4(v4f{accp[4], accp[5], accp[6], accp[7]});
    const v4f gg = tanh4(v4f{accp[8], accp[9], accp[10], accp[11]});
    const v4f og = sigmoid4(v4f{accp[12], accp[13], accp[14], accp[15]});
    const v4f cn = cell4(fg, creg[TP], ig, gg);
    const v4f hn = mul_rn4(og, tanh4(cn));
    creg[TP] = cn;
    const v4bfw hb4 = __builtin_convertvector(hn, v4bfw);
    if constexpr (ABL & 16) {
      creg[TP].x += (float)hb4[0];
      return;
    }
    if constexpr (NL > 0) {
      if (valid) {
        if constexpr (OUTF)
          bst4(hn, ro, voo, oo_t + 32 * TP);
        else
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, hb4), ro, voo, oo_t + 16 * TP, 0);
      }
    } else {
      pend_f = hn;
      pend_b = hb4;
      pend_off = oo_t + (OUTF ? 32 : 16) * TP;
      pend_live = true;
    }
    // stage h_t for the next step: operand block TP / 2, elements 4 (TP & 1) .. + 3
    *reinterpret_cast<v4bfw*>(hbuf + (TP / 2) * 1024 + 8 * (TP & 1)) = hb4;
  };
  auto flush_pending = [&]() {
    if constexpr (NL == 0) {
      if (pend_live && valid) {
        if constexpr (OUTF)
          bst4(pend_f, ro, voo, pend_off);
        else
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, pend_b), ro, voo, pend_off, 0);
      }
      pend_live = false;
    }
  };
  auto reload_h = [&]() {   // h_t becomes the recurrent operand (only this wave wrote / reads its staging area)
#pragma unroll
    for (int i = 0; i < NKH; ++i) hop[i] = __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(hbuf + i * 1024));
  };

  unsigned oo_prev = 0;
  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;
    const unsigned oo = tt * sto;

    static_for<NT>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if (T > 0 || step > 0) piece_barrier();               // see the protocol above
      flush_pending();
      // the next step's input blocks trickle in, one block per tile
      if constexpr (T < NKX && !(ABL & 32)) xn[T] = load_x(ic<T>{}, ttn);

      const char* cb = lds_rd + cslot * (KT * 1024);
      cslot = cslot + 1 == NSLOT ? 0 : cslot + 1;
      const char* nb_ = lds_rd + cslot * (KT * 1024);       // the next tile (already complete in LDS)
      // previous tile's gates, to be scheduled under this tile's MFMAs
      if constexpr (T > 0) {
        gate(ic<T - 1>{}, oo);
      } else {
        if (step > 0) gate(ic<NT - 1>{}, oo_prev);
      }
      // (two alternating accumulators were tried — a same-accumulator MFMA chain with instructions in between pays a
      //  re-issue penalty — and lost 8 %: the chain is not what paces this loop)
      v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // record K of tile T sits in register (T * KT + K) % AD of the A pipeline, which runs on across tiles;
      // NT * KT % AD == 0, so the phase only depends on the tile's position inside the step
      static_assert((NT * KT) % AD == 0, "A pipeline phase must repeat every step");
      static_for<KT>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr int R = (T * KT + K) % AD;
        if constexpr (K == KP) piece_barrier();             // second half of the tile
        if constexpr (T == 0 && K == 1 + NKX) {
          if (step > 0) reload_h();                         // the last tile's h has just been staged
        }
        if constexpr (!(ABL & 4)) {
          const v8bfw a = apipe[R];
          if constexpr (!(ABL & 8)) apipe[R] = (K + AD < KT) ? arec(cb, K + AD) : arec(nb_, K + AD - KT);
          if constexpr (K == 0)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc, 0, 0, 0);          // + bias (exact)
          else if constexpr (K <= NKX)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[K - 1 < NKX ? K - 1 : 0], acc, 0, 0, 0);
          else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hop[K - 1 - NKX], acc, 0, 0, 0);
        } else {
          acc[K % 16] += (float)apipe[R][0];
          apipe[R] = (K + AD < KT) ? arec(cb, K + AD) : arec(nb_, K + AD - KT);
        }
      });
      accp = acc;
    });
    static_for<(NKX > NT ? NKX - NT : 0)>([&](auto e) { xn[NT + decltype(e)::value] = load_x(ic<NT + decltype(e)::value>{}, ttn); });
#pragma unroll
    for (int i = 0; i < NKX; ++i) xb[i] = xn[i];
    oo_prev = oo;
  }
  // epilogue: the last tile of the last step
  flush_pending();
  gate(ic<NT - 1>{}, oo_prev);
  flush_pending();
  // drain: the look-ahead requested tiles past the end (harmless re-reads of the stream's first tiles)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int H, int NW, int NB0, int NB2, int FLAGS, int NSLOT, int NL, int ABL = 0>
int launch_bf16w_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  constexpr int KT = 1 + NB0 + NB2 + H / 16;
  const size_t lds = (size_t)NSLOT * KT * 1024 + (size_t)NW * (H / 16) * 1024;
  auto k = lstm_bf16w_kernel<H, NW, NB0, NB2, FLAGS, NSLOT, NL, ABL>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3((NW + NL) * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_bf16w_kernel");
  return FNSSL_OK;
}

// kNoStatic when the shape / element types have no instantiation
int launch_bf16w(const LstmParams& p, int H, int NW, int flags, int nwg, hipStream_t st);
int forward_bf16w(LstmParams p, int H, int flags, hipStream_t st, int* family = nullptr);

}  // namespace fnssl_lstm
