// Pair-split wide bf16 LSTM recurrence (BASELINE config 3): two waves share one group of 32 sequences.
//
// What the measurements of lstm_bf16w.h said (profiles/r02/f_ablate_bf16w.txt, config-3 narrow-band layer, one
// 32-sequence wave per SIMD on two of the four SIMDs, loader waves on the other two): 74 k cycles per step against
// 35 k of MFMA issue — and 56 k WITHOUT any MFMA.  The gate math (10 transcendentals + ~13 other VALU operations per
// cell, 128 cells per lane and step) costs more than the matrix product once the product runs on bf16 MFMAs, and with
// 64 sequences per CU it was concentrated on two SIMDs.  So the split that matters is the one that spreads BOTH:
//   * a workgroup = 2 groups x 2 roles; the two waves of a group take half of the gate-row tiles each (role r: tiles
//     r * NT/2 .. ), i.e. half of the MFMAs AND half of the gate math / cell state of the same 32 sequences, on four
//     SIMDs instead of two; every 1 KiB weight record is still read from LDS once per 32 sequences;
//   * h_t is exchanged inside the pair through the lane-private-by-construction LDS staging area lstm_bf16w.h
//     already used (8 bytes per tile and lane), with one of the ring's own barriers as the exchange point;
//   * the ring carries, per barrier interval, one QUARTER tile ("piece") for each role: slot = [role 0 | role 1],
//     6 slots (1 consumed, 1-2 landed, 3-4 in flight), filled by LDS-DMA; each wave requests every second record of
//     its own role's piece, so the four waves share the issue work;
//   * the gate math of tile i - 1 is issued under the MFMAs of tile i (software pipeline; its result is only needed
//     by the next step); the tile's h store is issued right after the following barrier.
// Where the time goes now (profiles/r02/f_ablate_bf16p.txt; config-3 narrow-band layer, 16384 sequences x 300 steps):
// 7.0 ms = 757 TFLOP/s (30 % of the bf16 MFMA roof; lstm_bf16.h: 11.1 ms).  Without the MFMAs 6.2 ms, without the
// transcendentals 6.5 ms, without both 4.45 ms: the floor is the weight stream — every CU pulls the whole 1.06 MB
// matrix from L2 once per step (84 GB per launch, 97.6 % L2 hits, 2.4 ms at the L2 peak), and with 64 sequences per
// CU (batch 64) there is nobody to share a pass with.  SQ counters: waves issue-active 62 %, MFMA pipes busy 36 %.
// Stream layout, operand permutation, bias-as-three-bf16-terms and element types: exactly those of lstm_bf16w.h
// (fnssl_lstm_pack_bf16w); only the ORDER in which tiles are fetched differs, and that is address arithmetic here.
#pragma once

#include "lstm_bf16w.h"

namespace fnssl_lstm {

// ABL (make ABLATE=1 builds only): timing ablations, wrong results — 1 cheap gate math (no transcendentals), 4 no MFMAs
// NG = 32-sequence groups per workgroup (2 roles each: 2 NG waves).  The ring, its barriers and the L2 weight stream
// are per WORKGROUP and step, so more groups per workgroup amortise them: the 400-column full-band layer of IPDnet has
// 1200 groups (64 utterances x 300 frames / 32, two directions) = 600 two-group workgroups = 2.34 rounds on 256 CUs run
// as 3; with NG = 5 it is 240 workgroups = ONE round (and the step's ring traffic serves 160 sequences instead of 64).
template <int H, int NB0, int NB2, int FLAGS, int ABL = 0, int NSLOT_ = 7, bool DRAIN = false, int NG = 2>
__global__ void __launch_bounds__(NG * 128) lstm_bf16p_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  constexpr int NT = H / 8, NTW = NT / 2, NKH = H / 16, NKX = NB0 + NB2, KT = 1 + NKX + NKH;
  constexpr int KP = KT / 2;                              // half a tile
  // ring granule = a QUARTER tile per role ("piece": KQ0, KP - KQ0, KQ0, KP - KQ0 records), one barrier per piece
  constexpr int KQ0 = (KP + 1) / 2, KQ1 = KP - KQ0;
  constexpr int KPW = (KQ0 + NG - 1) / NG;                // DMA requests per wave and piece (NG waves per role)
  // Ring accounting.  At the barrier that opens interval j every wave has REQUESTED pieces 0 .. j + NSLOT - 2 (the
  // prologue requests NSLOT - 1, one more follows each barrier) and waits until all but its INFL youngest piece
  // requests have landed, i.e. pieces <= j + LAND are complete in LDS for everybody after the barrier; the request
  // issued after the barrier (piece j + NSLOT - 1) takes the slot of interval j - 1, which nobody reads any more.
  //   slots: [j - 1: being refilled] [j: consumed] [j + 1 .. j + LAND: landed] [the INFL youngest: in flight]
  // The A-operand pipeline reads AD records ahead of the MFMA, so it may reach LAND pieces ahead and no further:
  // AD <= the smallest piece.  (Round 2 waited for vmcnt(INFL' * KPW) with INFL' = NSLOT - 1 - LAND counted AFTER the
  // next request, one piece too few: the read-ahead could touch a piece still in flight — stale weights under load.)
  constexpr int NSLOT = NSLOT_;                           // slots of KQ0 records per role
  constexpr int LAND = 1;                                 // pieces ahead of the consumer that are complete in LDS
  constexpr int INFL = NSLOT - 2 - LAND;                  // my piece requests that may still be in flight at a barrier
  constexpr bool F0 = FLAGS & kW_F0, F2 = FLAGS & kW_F2, OUTF = FLAGS & kW_OUTF;
  constexpr int AD = KQ1 >= 4 ? 4 : KQ1;                  // A-operand reads in flight ahead of their MFMA
  constexpr int SLOTB = 2 * KQ0 * 1024;                   // bytes per slot (both roles)
  constexpr int RING = NSLOT * SLOTB, HBUF = NKH * 1024;
  // the last tile's gates of a step run under the first tile of the next step when the recurrent part of a tile
  // starts after its mid-tile barrier (the pair's exchange point); otherwise they run at the end of their own step
  constexpr bool LATE = 1 + NKX >= KQ0;
  static_assert(KT % 2 == 0 && NT % 2 == 0, "tiles are fetched in two pieces, shared by two roles");
  static_assert(RING + NG * HBUF <= 160 * 1024, "ring + h staging do not fit the LDS");
  static_assert(AD >= 1 && AD <= KQ1 && KQ1 <= KQ0 && (NTW * KT) % AD == 0, "A pipeline reaches at most one piece ahead");
  static_assert(INFL >= 1 && NSLOT >= 6, "cbs[] addresses six consecutive slots");
  static_assert(INFL * KPW <= 60, "vmcnt is a 6-bit counter");
  static_assert(NG * (KPW - 1) - 1 < KQ1 || KPW == 1, "only the last request of a piece may need clamping");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int n = lane & 31, hb = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = w / NG, grp = w - role * NG;           // consumes tiles role * NTW + i of group grp
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int task = p.task0 + wg * NG + grp;               // 32-sequence group
  int q = task * 32 + n;
  const bool valid = q < p.nseq && task < p.task1;
  if (q >= p.nseq) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  constexpr int E0 = F0 ? 4 : 2, E2 = F2 ? 4 : 2, EO = OUTF ? 4 : 2;
  unsigned vo0 = 0, vo2 = 0, voo = 0;
  rsrc_t rx0 = make_rsrc(p.out), rx2 = make_rsrc(p.out);
  if constexpr (NB0 > 0) rx0 = split_addr_e<E0>(p.src0.p, qo * p.src0.so + qi * p.src0.si + 8 * hb, vo0);
  if constexpr (NB2 > 0) rx2 = split_addr_e<E2>(p.src2.p, qo * p.src2.so + qi * p.src2.si + 8 * hb, vo2);
  // my tiles start at unit 8 * role * NTW
  const rsrc_t ro = split_addr_e<EO>(p.out, qo * p.out_so + qi * p.out_si + dir * H + 8 * role * NTW + 4 * hb, voo);
  const unsigned st0 = (unsigned)(p.src0.st * E0), st2 = (unsigned)(p.src2.st * E2), sto = (unsigned)(p.out_st * EO);
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const unsigned lds0 = (unsigned)(size_t)smem;

  // ---- ring.  Piece j of a step (j = 4 i + quarter): role r consumes records [start(quarter), +size) of tile
  // r * NTW + i.  Wave w REQUESTS for its own role: records (w & 1), (w & 1) + 2, ... of the piece (a request past a
  // short piece repeats its last record: every wave issues exactly KPW requests per piece, which is what the
  // s_waitcnt vmcnt(INFL * KPW) accounting relies on — loads complete in order).
  // L2 is the wall here: every CU streams the whole matrix from L2 once per step (1.06 MB x 256 CUs x 300 steps =
  // 84 GB per launch = 2.4 ms at the 34.5 TB/s L2 peak), and a request waits ~1300 cycles when all 32 CUs of an XCD
  // pull the same piece at once — with one piece in flight the ring alone took 41 k cycles per step.
  const int fpar = grp;                                    // my index among the NG waves that fetch for this role
  const unsigned src_role = (unsigned)(role * NTW * KT) * 1024u;
  // (per-XCD tile rotation — the CUs of an XCD pulling DIFFERENT records at any moment — was measured in round 3: no gain)
  auto fetch_piece = [&](int j, int slot) {
    const int qt = j & 3;
    const int start = (qt >> 1) * KP + (qt & 1) * KQ0, size = (qt & 1) ? KQ1 : KQ0;
    const unsigned sb = src_role + (unsigned)((j >> 2) * KT + start + fpar) * 1024u;
    const unsigned lb = lds0 + (unsigned)(slot * SLOTB + (role * KQ0 + fpar) * 1024);
    // records fpar + 2 m: the first KPW - 1 always exist (2 (KPW - 2) + 1 < KQ1), the last one may run past a short
    // piece and then repeats the piece's last record
    static_for<KPW - 1>([&](auto mc) { dma16_imm<decltype(mc)::value * NG * 1024>(rw, vlane, sb, lb); });
    const int rl = fpar + NG * (KPW - 1) < size ? NG * (KPW - 1) : size - 1 - fpar;
    dma16_imm<0>(rw, vlane, sb + (unsigned)rl * 1024u, lb + (unsigned)rl * 1024u);
  };
  int fj = 0, fslot = 0;
  auto fetch_next = [&]() {
    fetch_piece(fj, fslot);
    fj = fj + 1 == 4 * NTW ? 0 : fj + 1;
    fslot = fslot + 1 == NSLOT ? 0 : fslot + 1;
  };
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) fetch_next();
  int cslot = 0;                                           // slot of the interval being consumed

  // ---- state ---------------------------------------------------------------------------------------------------
  const v4bfw zb4 = v4bfw{0, 0, 0, 0};
  v8bfw ones;
  {
    const __bf16 o1 = (__bf16)(hb == 0 ? 1.0f : 0.0f);
    ones = v8bfw{o1, o1, o1, 0, 0, 0, 0, 0};
  }
  v8bfw xb[NKX > 0 ? NKX : 1], xn[NKX > 0 ? NKX : 1];
  v8bfw hop[NKH];
  v4f creg[NTW];
#pragma unroll
  for (int i = 0; i < NKH; ++i) hop[i] = join8(zb4, zb4);
#pragma unroll
  for (int i = 0; i < NTW; ++i) creg[i] = v4f{0.f, 0.f, 0.f, 0.f};
  // the group's h_t staging area [block s][lane][16 B]: global tile 2 s -> bytes 0-7, 2 s + 1 -> 8-15
  char* const hbuf = smem + RING + grp * HBUF + lane * 16;

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

  // deferred h store: issued right after a barrier, so it is a whole interval old at the next vmcnt(0)
  v4f pend_f = {0.f, 0.f, 0.f, 0.f};
  v4bfw pend_b = zb4;
  unsigned pend_off = 0;
  bool pend_live = false;
  auto flush_pending = [&]() {
    if (pend_live && valid) {
      if constexpr (OUTF)
        bst4(pend_f, ro, voo, pend_off);
      else
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, pend_b), ro, voo, pend_off, 0);
    }
    pend_live = false;
  };
  // One barrier per interval: everything I requested so far has landed (my part of the NEXT interval's pieces);
  // barrier: so has everybody's, and everybody is done with the previous interval, whose slot the next request takes.
  // EXCH: the interval boundary at which the pair exchanges h_t — my staging writes must be complete first.
  auto piece_barrier = [&](auto exch) {
    if constexpr (decltype(exch)::value)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : INFL * KPW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DRAIN ? 0 : INFL * KPW) : "memory");
    flush_pending();
    fetch_next();
  };

  const char* const lds_rd = smem + lane * 16 + role * (KQ0 * 1024);  // my role's half of a slot
  auto arec = [&](const char* base, int r) { return __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(base + r * 1024)); };
  v16f accp = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto gate = [&](auto ic_, unsigned oo_t) {   // post-process my tile I (global tile role * NTW + I)
    constexpr int I = decltype(ic_)::value;
    v4f cn, hn;
    if constexpr (ABL & 1) {
      cn = creg[I] + v4f{accp[0], accp[5], accp[10], accp[15]};
      hn = cn * 0.5f + v4f{accp[1], accp[6], accp[11], accp[12]};
    } else {
      const v4f ig = sigmoid4(v4f{accp[0], accp[1], accp[2], accp[3]});
      const v4f fg = sigmoid4(v4f{accp[4], accp[5], accp[6], accp[7]});
      const v4f gg = tanh4(v4f{accp[8], accp[9], accp[10], accp[11]});
      const v4f og = sigmoid4(v4f{accp[12], accp[13], accp[14], accp[15]});
      cn = cell4(fg, creg[I], ig, gg);
      hn = mul_rn4(og, tanh4(cn));
    }
    creg[I] = cn;
    const v4bfw hb4 = __builtin_convertvector(hn, v4bfw);
    pend_f = hn;
    pend_b = hb4;
    const int ta = I;                                     // the stream tile of this static index
    pend_off = oo_t + (OUTF ? 32 : 16) * ta;
    pend_live = true;
    // global tile G = role * NTW + ta -> operand block G / 2, elements 4 (G & 1) .. + 3
    const int gt = role * NTW + ta;
    *reinterpret_cast<v4bfw*>(hbuf + (gt >> 1) * 1024 + 8 * (gt & 1)) = hb4;
  };
  auto reload_h = [&]() {
#pragma unroll
    for (int i = 0; i < NKH; ++i) hop[i] = __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(hbuf + i * 1024));
  };

  // the first interval: wait, publish, prime the A pipeline
  piece_barrier(ic<0>{});
  v8bfw apipe[AD];
#pragma unroll
  for (int i = 0; i < AD; ++i) apipe[i] = arec(lds_rd, i);

  unsigned oo_prev = 0;
  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;
    const unsigned oo = tt * sto;

    static_for<NTW>([&](auto tc) {
      constexpr int I = decltype(tc)::value;
      // !LATE shapes exchange h at the step's first barrier (the previous step finished all its gates)
      if (I > 0 || step > 0) {
        if constexpr (I == 0 && !LATE)
          piece_barrier(ic<1>{});
        else
          piece_barrier(ic<0>{});
      }
      if constexpr (I == 0 && !LATE) {
        if (step > 0) reload_h();
      }
      // The next step's input blocks are requested in ONE burst at the start of the step: vector loads return in
      // order, so a load that has to come from HBM holds back the weight requests queued behind it — once per step
      // instead of once per tile.  (Non-temporal hints on these loads and on the h stores were tried: 7.2 -> 9.8 ms.)
      if constexpr (I == 0) static_for<NKX>([&](auto b) { xn[decltype(b)::value] = load_x(b, ttn); });
      // slots of this tile's four pieces and of the next tile's first two
      const char* cbs[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const int sl = cslot + m < NSLOT ? cslot + m : cslot + m - NSLOT;
        cbs[m] = lds_rd + sl * SLOTB;
      }
      cslot = cslot + 4 < NSLOT ? cslot + 4 : cslot + 4 - NSLOT;
      if constexpr (I > 0) {
        gate(ic<I - 1>{}, oo);
      } else if constexpr (LATE) {
        if (step > 0) gate(ic<NTW - 1>{}, oo_prev);
      }
      v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      static_for<KT>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr int R = (I * KT + K) % AD;
        // piece boundaries inside the tile; the pair exchanges h_t at the last one before the recurrent blocks
        constexpr bool BND = K == KQ0 || K == KP || K == KP + KQ0;
        constexpr int KE = (1 + NKX >= KP + KQ0) ? KP + KQ0 : ((1 + NKX >= KP) ? KP : KQ0);
        if constexpr (BND) {
          if constexpr (I == 0 && LATE && K == KE)
            piece_barrier(ic<1>{});                         // the pair's h_t is complete in the staging area
          else
            piece_barrier(ic<0>{});
        }
        if constexpr (I == 0 && LATE && K == 1 + NKX) {
          if (step > 0) reload_h();
        }
        const v8bfw a = apipe[R];
        // record K + AD: piece index (0..3 this tile, 4..5 next tile) and offset inside the piece
        constexpr int KN = K + AD;
        constexpr int KM = KN % KT;
        constexpr int PC = (KN / KT) * 4 + (KM >= KP ? 2 : 0) + ((KM % KP) >= KQ0 ? 1 : 0);
        constexpr int OFF = (KM % KP) >= KQ0 ? (KM % KP) - KQ0 : (KM % KP);
        apipe[R] = arec(cbs[PC], OFF);
        if constexpr (ABL & 4) {
          acc[K % 16] += (float)a[0];
        } else if constexpr (K == 0) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc, 0, 0, 0);          // + bias (exact)
        } else if constexpr (K <= NKX) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[K - 1 < NKX ? K - 1 : 0], acc, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hop[K - 1 - NKX], acc, 0, 0, 0);
        }
      });
      accp = acc;
    });
    if constexpr (!LATE) gate(ic<NTW - 1>{}, oo);
#pragma unroll
    for (int i = 0; i < NKX; ++i) xb[i] = xn[i];
    oo_prev = oo;
  }
  if constexpr (LATE) {
    flush_pending();
    gate(ic<NTW - 1>{}, oo_prev);
  }
  flush_pending();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ALLDRAIN: the drained barriers for every shape (FNSSL_BF16P_DRAIN=1; a diagnosis knob: if the stale-record effect below ever
// shows up on a shape that owns its CU, this tells ring accounting from residency)
template <int H, int NB0, int NB2, int FLAGS, int ABL = 0, int NG = 2, bool ALLDRAIN = false>
int launch_bf16p_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  constexpr int KT = 1 + NB0 + NB2 + H / 16;
  // 7 slots where a workgroup owns its CU anyway; the 144-column full-band layer (6 KB slots) keeps 6 so that several
  // workgroups stay resident per CU
  constexpr int NSLOT = KT <= 12 ? 6 : 7;
  const size_t lds = (size_t)NSLOT * 2 * ((KT / 2 + 1) / 2) * 1024 + (size_t)NG * (H / 16) * 1024;
  // DRAIN (the 144-column layer only, the one shape whose workgroups share a CU): every barrier
  // waits for ALL of the wave's requests (vmcnt(0)) instead of leaving INFL pieces in flight.  Measured on MI355X
  // (profiles/r03/c_*): with counted waits this shape alone gave run-to-run differences at config 3's batch (a 32-sequence
  // group reading one stale weight record, a few times per launch) whenever several workgroups were resident per CU
  // — bit-stable with one workgroup per CU (same kernel, LDS padded) and with drained barriers; the cause could not
  // be pinned from here, so the shape runs the canonical "vmcnt(0) + barrier" protocol: 3.99 -> 4.37 ms per launch
  // (three resident workgroups hide each other's waits).  The other shapes own their CU (114-158 KB of LDS).
  auto k = lstm_bf16p_kernel<H, NB0, NB2, FLAGS, ABL, NSLOT, (KT <= 12) || ALLDRAIN, NG>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NG * 128), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_bf16p_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
