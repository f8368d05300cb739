// Operand-ring kernel for the H = 256 narrow-band layers, FOUR hidden slices per pass (round 6): lstm_static3_kernel with
// half the passes over [x_t | h_{t-1}] and the weight ring staged by LDS-DMA instead of through registers.
//
// Why.  lstm_static3_kernel walks the 32 operand blocks of a step once per PAIR of hidden slices: 8 passes, 16 KiB of
// operand re-reads per sequence-step against 7 KiB of algorithmic traffic — measured 3.0 x the algorithmic bytes at the
// fabric side of L2 (profiles/r05/hbm_traffic.json; the re-reads are Infinity-Cache hits, the working set of a step is
// 12 MB per XCD against 4 MB of L2).  Four slices per pass halve the re-reads (4 passes, 8 KiB) and the operand-load
// instructions beside the matrix stream.  They need 32 more accumulator registers than lstm_static3_kernel's 146 of the
// 168 that three waves per SIMD allow; its 16 staging registers (the weight chunk on its way L2 -> registers -> LDS) are
// what pays for them: here every wave requests its share of the NEXT chunk straight into the ring slot with
// `buffer_load_dwordx4 ... lds` right after the barrier that frees the slot, and certifies it with a counted
// `s_waitcnt vmcnt(K)` in front of the next barrier — K = the operand-block loads the wave has issued since (known at
// compile time per chunk; anything else it issued in between only makes the wait stricter; vector memory operations
// complete in order).
//
// Weight stream: QUAD-interleaved [slice quad][quad][slice in quad][4 records] (quad_stream_kernel, lstm.hip: one tiny
// launch per call into the workspace, like the pair-interleaved copy of lstm_static3_kernel).  Same k order per sequence
// and slice, same gate code: bit-identical to every other fp32 family.
#pragma once

#include "lstm_static.h"

#pragma clang fp contract(off)

namespace fnssl_lstm {

// block quads among the CHQ virtual quads of the chunk that ends with virtual quad QE: quad 0 is the bias quad, quads
// 1 .. NV0 the input blocks, quad 1 + NV0 the 4-channel remainder (NS2), then NS recurrent blocks, then padding
constexpr int static4_blocks_in_chunk(int qe, int chq, int nv0, int ns2, int ns) {
  int n = 0;
  for (int q = qe - chq + 1; q <= qe; ++q) {
    const bool x = q >= 1 && q <= nv0;
    const bool h = q >= 1 + nv0 + ns2 && q < 1 + nv0 + ns2 + ns;
    n += (x || h) ? 1 : 0;
  }
  return n;
}

// PF4: the cell state and residual operand of ALL four slices are requested at the start of the pass's last chunk (24 registers
// the 256-channel variants have to spare) instead of slice s + 1's under slice s's gate math (one L2 / HBM round trip exposed
// per slice, three per pass) — block 1's variant (166 of 168 registers) keeps the one-ahead form.
template <int H, int NW, int NV0, int NS2, int CHQ, int PAD, int MODE, int XD = 4, bool PF4 = false>
__global__ void __launch_bounds__(NW * 64) lstm_static4_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  constexpr int NS = H / 16, NP = NS / 4;
  constexpr int NB = NV0 + NS;                          // sixteen-channel operand blocks per pass: x_t, then h_{t-1}
  constexpr bool HAS2 = (MODE & kHas2) != 0, SUM = (MODE & kSum) != 0;
  static_assert(!(MODE & kHas1) && HAS2 == (NS2 > 0) && NS % 4 == 0 && NS2 <= 1, "modes");
  constexpr int QPS = 1 + NV0 + NS2 + NS;               // real quads per slice
  constexpr int VQ = QPS + PAD;
  static_assert(VQ % CHQ == 0, "chunks must tile the (padded) slice quad");
  constexpr int CH = 16 * CHQ;                          // records per chunk: CHQ quads x 4 slices x 4 records
  constexpr int M = (CH + NW - 1) / NW;                 // DMA instructions per wave and chunk
  static_assert(NB % XD == 0 && XD <= NV0, "the operand ring depth must divide the block count");
  static_assert(PAD < CHQ, "a chunk of padding only would have no operand load to count the DMA by");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int task = p.task0 + wg * NW + w;
  int q = task * 16 + n;
  const bool valid = q < p.nseq && task < p.task1;
  if (q >= p.nseq) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  unsigned vo0 = 0, vo2 = 0, voo = 0, vok = 0, voo2 = 0;
  const rsrc_t rx0 = split_addr(p.src0.p, qo * p.src0.so + qi * p.src0.si, 4 * g, vo0);
  const rsrc_t rx2 = HAS2 ? split_addr(p.src2.p, qo * p.src2.so + qi * p.src2.si, g, vo2) : rx0;
  const rsrc_t rsk = SUM ? split_addr(p.skip.p, qo * p.skip.so + qi * p.skip.si, dir * H + 4 * g, vok) : rx0;
  const rsrc_t ro = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo);
  const rsrc_t ro2 = SUM ? split_addr(p.out_sum, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo2) : ro;
  // h_{-1} = 0: the same base with ZERO records — every lane is out of range and the load returns 0
  const rsrc_t rzero = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.out), 0, 0, 0x00020000);
  const rsrc_t rc = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                              ((size_t)dir * (p.ntasks + 16) + (task < p.task1 ? task : p.ntasks + w)) * (NS * 1024));
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned st0 = (unsigned)(p.src0.st * 4), st2 = HAS2 ? (unsigned)(p.src2.st * 4) : 0u;
  const unsigned sto = (unsigned)(p.out_st * 4), stk = SUM ? (unsigned)(p.skip.st * 4) : 0u;
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;

  // ---- weight ring: 2 slots of CHQ quads = 16 CHQ records, filled by LDS-DMA (1 KiB per wave-instruction: lane l's 16
  // bytes land at the slot address + 16 l; not counted by the compiler — see the vmcnt in ring_end) ---------------------
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  char* const lds_rd = smem + lane * 16;
  int dslot = 0, rslot = 0;
  int src_rec = 0;        // record index of the next chunk to request (quad-interleaved stream)
  int src_vq = 0;         // its virtual quad offset inside the slice quad
  auto issue_dma = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int r = w + m * NW;
      if (r < CH && src_vq * 16 + r < QPS * 16) {
        const unsigned soff = (unsigned)(src_rec + r) * 1024u;
        const unsigned ld = lds0 + (unsigned)(dslot * CH + r) * 1024u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(vlane), "s"(rw),
                     "s"(__builtin_amdgcn_readfirstlane(ld)), "s"(__builtin_amdgcn_readfirstlane(soff))
                     : "memory");
      }
    }
    dslot ^= 1;
    src_vq += CHQ;
    src_rec += CH;
    if (src_vq == VQ) {
      src_vq = 0;
      src_rec -= PAD * 16;                      // the padding quads do not exist in the stream
      if (src_rec == NP * QPS * 16) src_rec = 0;
    }
  };
  issue_dma();                                  // chunk 0 -> slot 0
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  issue_dma();                                  // chunk 1 -> slot 1, lands under chunk 0
  const char* cb = lds_rd;
  auto rec = [&](auto ql, int j) { return *reinterpret_cast<const v4f*>(cb + decltype(ql)::value * 16384 + j * 1024); };
  v4f a0 = rec(ic<0>{}, 0), a1 = rec(ic<0>{}, 1);

  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f acc[4][4];                                // [slice of the quad][gate]

  auto ring_step = [&](auto qi_c) {
    constexpr int QL = decltype(qi_c)::value % CHQ;
    if constexpr (QL + 1 < CHQ) {
      a0 = rec(ic<QL + 1>{}, 0);
      a1 = rec(ic<QL + 1>{}, 1);
    }
  };
  auto ring_end = [&](auto qi_c) {
    constexpr int QI = decltype(qi_c)::value, QL = QI % CHQ;
    if constexpr (QL + 1 == CHQ) {
      // my share of the next chunk has landed: everything but the K operand-block loads issued since its request is complete
      constexpr int K = static4_blocks_in_chunk(QI, CHQ, NV0, NS2, NS);
#ifdef FNSSL_BUILD_ABLATE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ablation twins drop operand loads: no count to rely on)
#else
      if constexpr (K >= 3)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr (K == 2)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if constexpr (K == 1)
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everybody's has, and everybody is done with this slot
      issue_dma();                                                      // the chunk after the next, into the slot just freed
      rslot ^= 1;
      cb = lds_rd + rslot * (CH * 1024);
      a0 = rec(ic<0>{}, 0);
      a1 = rec(ic<0>{}, 1);
    }
  };
  // one operand block against the four slices: 16 records, two read ahead of the MFMAs that use them
#define SQUAD4(QI, B0, B1, B2, B3)                                                \
  do {                                                                            \
    const v4f a2_ = rec(ic<(QI) % CHQ>{}, 2), a3_ = rec(ic<(QI) % CHQ>{}, 3);     \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[0], a0, B0);                                                        \
    MFMA4(acc[0], a1, B1);                                                        \
    const v4f b0_ = rec(ic<(QI) % CHQ>{}, 4), b1_ = rec(ic<(QI) % CHQ>{}, 5);     \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[0], a2_, B2);                                                       \
    MFMA4(acc[0], a3_, B3);                                                       \
    const v4f b2_ = rec(ic<(QI) % CHQ>{}, 6), b3_ = rec(ic<(QI) % CHQ>{}, 7);     \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[1], b0_, B0);                                                       \
    MFMA4(acc[1], b1_, B1);                                                       \
    const v4f c0_ = rec(ic<(QI) % CHQ>{}, 8), c1_ = rec(ic<(QI) % CHQ>{}, 9);     \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[1], b2_, B2);                                                       \
    MFMA4(acc[1], b3_, B3);                                                       \
    const v4f c2_ = rec(ic<(QI) % CHQ>{}, 10), c3_ = rec(ic<(QI) % CHQ>{}, 11);   \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[2], c0_, B0);                                                       \
    MFMA4(acc[2], c1_, B1);                                                       \
    const v4f d0_ = rec(ic<(QI) % CHQ>{}, 12), d1_ = rec(ic<(QI) % CHQ>{}, 13);   \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[2], c2_, B2);                                                       \
    MFMA4(acc[2], c3_, B3);                                                       \
    const v4f d2_ = rec(ic<(QI) % CHQ>{}, 14), d3_ = rec(ic<(QI) % CHQ>{}, 15);   \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[3], d0_, B0);                                                       \
    MFMA4(acc[3], d1_, B1);                                                       \
    ring_step(ic<(QI)>{});                                                        \
    __builtin_amdgcn_sched_barrier(0);                                            \
    MFMA4(acc[3], d2_, B2);                                                       \
    MFMA4(acc[3], d3_, B3);                                                       \
    ring_end(ic<(QI)>{});                                                         \
  } while (0)
#define SQUAD4_1(QI, B0)                                                                          \
  do {                                                                                            \
    const v4f b0_ = rec(ic<(QI) % CHQ>{}, 4), c0_ = rec(ic<(QI) % CHQ>{}, 8), d0_ = rec(ic<(QI) % CHQ>{}, 12); \
    MFMA4(acc[0], a0, B0);                                                                        \
    MFMA4(acc[1], b0_, B0);                                                                       \
    MFMA4(acc[2], c0_, B0);                                                                       \
    MFMA4(acc[3], d0_, B0);                                                                       \
    ring_step(ic<(QI)>{});                                                                        \
    ring_end(ic<(QI)>{});                                                                         \
  } while (0)

  // ---- operand ring: block b of a pass (b < NV0: channels 16 b.. of x_t; else hidden units 16 (b - NV0).. of h_{t-1})
  // lives in br[b % XD] and is requested right after block b - XD has been consumed
  v4f br[XD];
  {
    const unsigned tt0 = rev ? p.nsteps - 1 : 0;
    static_for<XD>([&](auto v) { br[v.value] = bld4(rx0, vo0, tt0 * st0 + 64 * v.value); });
  }

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;
    const unsigned o0 = tt * st0, o2 = tt * st2, oo = tt * sto, ok = tt * stk;
    const unsigned op = (rev ? tt + 1 : tt - 1) * sto;     // row of h_{step - 1} (not addressed at step 0: zero records)
    const rsrc_t rh = step > 0 ? ro : rzero;
    float xs2 = 0.f;
    if (NS2) xs2 = bld1(rx2, vo2, o2);

    for (int pr = 0; pr < NP; ++pr) {
      const int s0 = 4 * pr;
#ifdef FNSSL_BUILD_ABLATE   // timing ablation (wrong results): FNSSL_STATIC3_ABL bit 2 = no operand loads at all
      const bool skip_ring = (p.ablate & 2) != 0;
#else
      constexpr bool skip_ring = false;
#endif
      v4f cprev = zero4, skipv = zero4;
      v4f cpf[PF4 ? 4 : 1], skf[PF4 ? 4 : 1];
      if constexpr (PF4) static_for<4>([&](auto s_) { cpf[s_.value] = zero4; skf[s_.value] = zero4; });
      const bool last = pr + 1 == NP;
      // the input blocks requested across the end of the pass: the same row, or (last quad) the next step's
      const unsigned nx = (last ? ttn : tt) * st0;
      // quad 0: the four slices' bias records -> accumulators
      acc[0][0] = a0;
      acc[0][1] = a1;
      static_for<14>([&](auto j) {
        constexpr int J = decltype(j)::value + 2;
        acc[J / 4][J % 4] = rec(ic<0>{}, J);
      });
      ring_step(ic<0>{});
      ring_end(ic<0>{});
      static_for<NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        constexpr int QI = B < NV0 ? 1 + B : 1 + NS2 + B;          // the 4-channel quad of block 1 sits between x and h
        if constexpr (NS2 > 0 && B == NV0) SQUAD4_1(1 + NV0, xs2);
        const v4f ob = br[B % XD];
        // request block B + XD (of this pass, or — wrapping — of the next one) BEFORE this block's chunk can end: the
        // count in ring_end relies on one operand-block load per block quad having been issued ahead of its barrier
        constexpr int BN = (B + XD) % NB;
        if (!skip_ring) {
          if constexpr (B + XD < NB) {
            if constexpr (BN < NV0)
              br[B % XD] = bld4(rx0, vo0, o0 + 64 * BN);
            else
              br[B % XD] = bld4(rh, voo, op + 64 * (BN - NV0));
          } else {
            static_assert(BN < NV0, "the wrapped requests are input blocks");
            br[B % XD] = bld4(rx0, vo0, nx + 64 * BN);
          }
        }
        if constexpr (!PF4 && B == NB - 4) {   // cell state / residual operand of the first slice: four quads ahead of their use
          if (step > 0) cprev = bld4(rc, vlane, s0 * 1024);
          if (SUM) skipv = bld4(rsk, vok, ok + 64 * s0);
        }
        if constexpr (PF4 && B == NB - 3) {    // ... of all four slices, at the start of the pass's last chunk
          static_for<4>([&](auto s_) {
            constexpr int S = decltype(s_)::value;
            if (step > 0) cpf[S] = bld4(rc, vlane, (s0 + S) * 1024);
            if (SUM) skf[S] = bld4(rsk, vok, ok + 64 * (s0 + S));
          });
        }
        SQUAD4(QI, ob.x, ob.y, ob.z, ob.w);
      });
      static_for<PAD>([&](auto u) {
        ring_step(ic<QPS + decltype(u)::value>{});
        ring_end(ic<QPS + decltype(u)::value>{});
      });
      // cell updates of the four slices; the next slice's cell state / residual operand in flight under this one's gate math
      static_for<4>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        const v4f ig = sigmoid4(acc[S][0]), fg = sigmoid4(acc[S][1]), gg = tanh4(acc[S][2]), og = sigmoid4(acc[S][3]);
        v4f cp_cur = cprev, sk_cur = skipv;
        if constexpr (PF4) {
          cp_cur = cpf[S];
          sk_cur = skf[S];
        } else if constexpr (S < 3) {
          if (step > 0) cprev = bld4(rc, vlane, (s0 + S + 1) * 1024);
          if (SUM) skipv = bld4(rsk, vok, ok + 64 * (s0 + S + 1));
        }
        const v4f cn = cell4(fg, cp_cur, ig, gg);
        v4f hn = mul_rn4(og, tanh4(cn));
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));   // h + skip adds the ROUNDED h
        bst4(cn, rc, vlane, (s0 + S) * 1024);
        if (valid) {
          bst4(hn, ro, voo, oo + 64 * (s0 + S));
          if (SUM) bst4(add_rn4(hn, sk_cur), ro2, voo2, oo + 64 * (s0 + S));
        }
      });
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (two chunks of DMA are still in flight towards this workgroup's LDS)
#undef SQUAD4
#undef SQUAD4_1
}

template <int H, int NW, int NV0, int NS2, int CHQ, int PAD, int MODE, int XD = 4, bool PF4 = false>
int launch_static4_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = (size_t)2 * CHQ * 16384;
  static_assert(2 * CHQ * 16384 <= 160 * 1024, "ring does not fit the LDS");
  auto k = lstm_static4_kernel<H, NW, NV0, NS2, CHQ, PAD, MODE, XD, PF4>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_static4_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
