// Operand-ring kernel for the H = 256 narrow-band layers (round 4): lstm_static2_kernel with the recurrent operand h_{t-1}
// STREAMED like the input operand x_t instead of held in 64 registers for the whole step.
//
// Why.  lstm_static2_kernel needs 168 of the 168 registers three waves per SIMD allow — 64 of them for h_{t-1} — and its
// register demand peaks at the cell update; the allocator spilled the per-lane offsets of the residual operands to scratch
// memory, and each reload (scratch_load + s_waitcnt vmcnt(0)) drained the wave's whole prefetch window three times per
// slice pair: the residual-summing variant ran 3.3 % behind the plain one (112.2 against 108.5 ms per launch at config 2,
// profiles/r04/) for two extra memory operations per slice.  h_t is written to the output tensor anyway and every step
// already re-read it from there once; here each slice PAIR re-reads it through the same 4-deep operand ring as x_t:
// [x_t | h_{t-1}] is one stream of 32 sixteen-channel blocks per pass.  That is 16 more loads per pass (measured cost of a
// load beside the matrix instructions: ~13 cycles, tools/ubench/issue_model.hip: +0.6 %), and frees 48 registers: no
// spills (146 registers), and room for the concatenated 4-channel input of block 1, which had to stay on the one-slice
// kernel.  Measured at config 2 (profiles/r04/): 110.3 against 112.0 ms per launch with the fused residual, 107.5 against
// 109.2 without, 110.7 against 111.1 for block 1's layer.
//
// Same weight stream (pair-interleaved), same k order per sequence and slice, same gate code: bit-identical results.
// Step 0 reads h_{-1} = 0 through a descriptor with zero records (out-of-range buffer loads return 0): no branch around
// the loads.  The h row of step t - 1 is complete before step t starts (this wave wrote it; a wave's own accesses to one
// address stay ordered), exactly as the once-per-step reload of lstm_static2_kernel assumed.
#pragma once

#include "lstm_static.h"

#pragma clang fp contract(off)

namespace fnssl_lstm {

template <int H, int NW, int M, int NV0, int NS2, int CHQ, int PAD, int MODE, int XD = 4>
__global__ void __launch_bounds__(NW * 64) lstm_static3_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  constexpr int NS = H / 16, NP = NS / 2;
  constexpr int NB = NV0 + NS;                          // sixteen-channel operand blocks per pass: x_t, then h_{t-1}
  constexpr bool HAS2 = (MODE & kHas2) != 0, SUM = (MODE & kSum) != 0;
  static_assert(!(MODE & kHas1) && HAS2 == (NS2 > 0) && NS % 2 == 0 && NS2 <= 1, "modes");
  constexpr int QPS = 1 + NV0 + NS2 + NS;               // real (pair-)quads per slice pair
  constexpr int VQ = QPS + PAD;
  static_assert(VQ % CHQ == 0, "chunks must tile the (padded) slice pair");
  constexpr int CH = 8 * CHQ;                           // records per chunk
  static_assert(CH <= NW * M, "chunk does not fit the staging registers");
  static_assert(NB % XD == 0 && XD <= NV0, "the operand ring depth must divide the block count");
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

  // ---- weight ring (2 slots of CHQ pair-quads = 8 CHQ records), as lstm_static2_kernel ------------------------------
  char* const lds_rd = smem + lane * 16;
  char* const lds_wr = smem + w * 1024 + lane * 16;
  int wslot = 0, rslot = 0;
  int src_rec = 0;        // record index of the next chunk to stage (pair-interleaved stream)
  int src_vq = 0;         // its virtual pair-quad offset inside the slice pair
  v4f stg[M];
  auto issue_loads = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int r = w + m * NW;
      if (r < CH && src_vq * 8 + r < QPS * 8) stg[m] = bld4(rw, vlane, (unsigned)(src_rec + r) * 1024u);
    }
    src_vq += CHQ;
    src_rec += CH;
    if (src_vq == VQ) {
      src_vq = 0;
      src_rec -= PAD * 8;                       // the padding quads do not exist in the stream
      if (src_rec == NP * QPS * 8) src_rec = 0;
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (w + m * NW < CH) *reinterpret_cast<v4f*>(lds_wr + wslot * (CH * 1024) + m * (NW * 1024)) = stg[m];
    wslot ^= 1;
  };
  auto sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  issue_loads();
  stage_write();
  sync();
  issue_loads();
  const char* cb = lds_rd;
  auto rec = [&](auto ql, int j) { return *reinterpret_cast<const v4f*>(cb + decltype(ql)::value * 8192 + j * 1024); };
  v4f a0 = rec(ic<0>{}, 0), a1 = rec(ic<0>{}, 1);

  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f acc[4], acd[4];                              // the pair's first / second slice

  auto ring_step = [&](auto qi_c) {
    constexpr int QL = decltype(qi_c)::value % CHQ;
    if constexpr (QL + 1 < CHQ) {
      a0 = rec(ic<QL + 1>{}, 0);
      a1 = rec(ic<QL + 1>{}, 1);
    }
  };
  auto ring_end = [&](auto qi_c) {
    constexpr int QL = decltype(qi_c)::value % CHQ;
    if constexpr (QL + 1 == (CHQ + 1) / 2 && CHQ > 1) stage_write();
    if constexpr (QL + 1 == CHQ) {
      if constexpr (CHQ == 1) stage_write();
      sync();
      issue_loads();
      rslot ^= 1;
      cb = lds_rd + rslot * (CH * 1024);
      a0 = rec(ic<0>{}, 0);
      a1 = rec(ic<0>{}, 1);
    }
  };
#define SQUAD3(QI, B0, B1, B2, B3)                                              \
  do {                                                                          \
    const v4f a2_ = rec(ic<(QI) % CHQ>{}, 2), a3_ = rec(ic<(QI) % CHQ>{}, 3);   \
    __builtin_amdgcn_sched_barrier(0);                                          \
    MFMA4(acc, a0, B0);                                                         \
    MFMA4(acc, a1, B1);                                                         \
    const v4f b0_ = rec(ic<(QI) % CHQ>{}, 4), b1_ = rec(ic<(QI) % CHQ>{}, 5);   \
    __builtin_amdgcn_sched_barrier(0);                                          \
    MFMA4(acc, a2_, B2);                                                        \
    MFMA4(acc, a3_, B3);                                                        \
    const v4f b2_ = rec(ic<(QI) % CHQ>{}, 6), b3_ = rec(ic<(QI) % CHQ>{}, 7);   \
    __builtin_amdgcn_sched_barrier(0);                                          \
    MFMA4(acd, b0_, B0);                                                        \
    MFMA4(acd, b1_, B1);                                                        \
    ring_step(ic<(QI)>{});                                                      \
    __builtin_amdgcn_sched_barrier(0);                                          \
    MFMA4(acd, b2_, B2);                                                        \
    MFMA4(acd, b3_, B3);                                                        \
    ring_end(ic<(QI)>{});                                                       \
  } while (0)
#define SQUAD3_1(QI, B0)                                      \
  do {                                                        \
    const v4f b0_ = rec(ic<(QI) % CHQ>{}, 4);                 \
    MFMA4(acc, a0, B0);                                       \
    MFMA4(acd, b0_, B0);                                      \
    ring_step(ic<(QI)>{});                                    \
    ring_end(ic<(QI)>{});                                     \
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
      const int s0 = 2 * pr;
#ifdef FNSSL_BUILD_ABLATE   // timing ablation (wrong results): FNSSL_STATIC3_ABL bit 1 = no operand loads in every second pass (what
      // four slices per pass would request), bit 2 = no operand loads at all
      const bool skip_ring = (p.ablate & 2) || ((p.ablate & 1) && (pr & 1));
#else
      constexpr bool skip_ring = false;
#endif
      v4f cprev0 = zero4, cprev1 = zero4, skip0 = zero4, skip1 = zero4;
      const bool last = pr + 1 == NP;
      // the input blocks requested across the end of the pass: the same row, or (last pair) the next step's
      const unsigned nx = (last ? ttn : tt) * st0;
      // quad 0: the two slices' bias records -> accumulators
      acc[0] = a0;
      acc[1] = a1;
      acc[2] = rec(ic<0>{}, 2);
      acc[3] = rec(ic<0>{}, 3);
      acd[0] = rec(ic<0>{}, 4);
      acd[1] = rec(ic<0>{}, 5);
      acd[2] = rec(ic<0>{}, 6);
      acd[3] = rec(ic<0>{}, 7);
      ring_step(ic<0>{});
      ring_end(ic<0>{});
      static_for<NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        constexpr int QI = B < NV0 ? 1 + B : 1 + NS2 + B;          // the 4-channel quad of block 1 sits between x and h
        if constexpr (NS2 > 0 && B == NV0) SQUAD3_1(1 + NV0, xs2);
        const v4f ob = br[B % XD];
        SQUAD3(QI, ob.x, ob.y, ob.z, ob.w);
        // request block B + XD: of this pass, or (wrapping) of the next one
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
        if constexpr (B == NB - 4) {       // cell state / residual operand of the first slice: four quads ahead of their use
          if (step > 0) cprev0 = bld4(rc, vlane, s0 * 1024);
          if (SUM) skip0 = bld4(rsk, vok, ok + 64 * s0);
        }
      });
      static_for<PAD>([&](auto u) {
        ring_step(ic<QPS + decltype(u)::value>{});
        ring_end(ic<QPS + decltype(u)::value>{});
      });
      // cell updates of the two slices
      {
        const v4f ig = sigmoid4(acc[0]), fg = sigmoid4(acc[1]), gg = tanh4(acc[2]), og = sigmoid4(acc[3]);
        if (step > 0) cprev1 = bld4(rc, vlane, s0 * 1024 + 1024);   // in flight under the first slice's gate math
        if (SUM) skip1 = bld4(rsk, vok, ok + 64 * s0 + 64);
        const v4f cn = cell4(fg, cprev0, ig, gg);
        v4f hn = mul_rn4(og, tanh4(cn));
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));   // h + skip adds the ROUNDED h
        bst4(cn, rc, vlane, s0 * 1024);
        if (valid) {
          bst4(hn, ro, voo, oo + 64 * s0);
          if (SUM) bst4(add_rn4(hn, skip0), ro2, voo2, oo + 64 * s0);
        }
      }
      {
        const v4f ig = sigmoid4(acd[0]), fg = sigmoid4(acd[1]), gg = tanh4(acd[2]), og = sigmoid4(acd[3]);
        const v4f cn = cell4(fg, cprev1, ig, gg);
        v4f hn = mul_rn4(og, tanh4(cn));
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));
        bst4(cn, rc, vlane, s0 * 1024 + 1024);
        if (valid) {
          bst4(hn, ro, voo, oo + 64 * s0 + 64);
          if (SUM) bst4(add_rn4(hn, skip1), ro2, voo2, oo + 64 * s0 + 64);
        }
      }
    }
  }
#undef SQUAD3
#undef SQUAD3_1
}

template <int H, int NW, int M, int NV0, int NS2, int CHQ, int PAD, int MODE, int XD = 4>
int launch_static3_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = (size_t)2 * CHQ * 8192;
  static_assert(2 * CHQ * 8192 <= 160 * 1024, "ring does not fit the LDS");
  auto k = lstm_static3_kernel<H, NW, M, NV0, NS2, CHQ, PAD, MODE, XD>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_static3_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
