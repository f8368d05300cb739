// Two hidden slices per pass: twin of lstm_static_kernel (lstm_static.h) for the H = 256 narrow-band layers at 12 waves
// per workgroup.  Same formulation and the same k-ordered fp32 MFMA chain per sequence and slice (bit-identical results);
// what changes is the LOOP ORDER: a pass multiplies every k-quad of [x_t | h_{t-1}] into the accumulators of TWO slices
// (32 output units), so each B operand — in particular each x_t block fetched from memory — is used for 32 MFMAs
// instead of 16 and x_t is re-read 8 times per step instead of 16 (the re-reads were 6.5 % of the kernel and 2/3 of its
// HBM-side traffic: profiles/r03/hbm_traffic_lstm_h256.json).  The weight stream must then deliver the two slices' records
// of a k-quad together: the "pair-interleaved" order [slice pair][quad][slice in pair][4 records], produced from the
// standard stream by a permutation of its 1-KiB records (fnssl_lstm_pack_pairs).  A ring "quad" is therefore 8 records.
#pragma once

#include "lstm_static.h"

#pragma clang fp contract(off)

#ifndef FNSSL_STATIC2_GPK
#define FNSSL_STATIC2_GPK true
#endif
#ifndef FNSSL_STATIC2_GPK2
#define FNSSL_STATIC2_GPK2 false
#endif

namespace fnssl_lstm {

// ABL = true: timing-ablation twin (make ABLATE=1 only, wrong results), bits of FNSSL_ABLATE: 1 no x loads, 2 cheap gates,
// 4 no stores, 8 no ring barrier, 16 no c / skip loads, 32 no h reload, 64 no LDS record reads, 128 no weight staging
template <int H, int NW, int M, int NV0, int NS2, int CHQ, int PAD, int MODE, int XD = 4, bool ABL = false>
__global__ void __launch_bounds__(NW * 64) lstm_static2_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  constexpr int NS = H / 16, NP = NS / 2;
  constexpr bool GPK2 = FNSSL_STATIC2_GPK2;
  constexpr bool GPK = FNSSL_STATIC2_GPK;   // packed gate math (lstm_kernel.h): only if the register allocation takes it without spilling
  constexpr bool HAS2 = (MODE & kHas2) != 0, SUM = (MODE & kSum) != 0;
  static_assert(!(MODE & kHas1) && HAS2 == (NS2 > 0) && NS % 2 == 0, "modes");
  constexpr int QPS = 1 + NV0 + NS2 + NS;               // real (pair-)quads per slice pair
  constexpr int VQ = QPS + PAD;
  static_assert(VQ % CHQ == 0, "chunks must tile the (padded) slice pair");
  constexpr int CH = 8 * CHQ;                           // records per chunk
  static_assert(CH <= NW * M, "chunk does not fit the staging registers");
  static_assert(NV0 % XD == 0, "the x ring depth must divide the block count");
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
  const rsrc_t rc = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                              ((size_t)dir * (p.ntasks + 16) + (task < p.task1 ? task : p.ntasks + w)) * (NS * 1024));
  const unsigned cy = (unsigned)p.carry;
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned st0 = (unsigned)(p.src0.st * 4), st2 = HAS2 ? (unsigned)(p.src2.st * 4) : 0u;
  const unsigned sto = (unsigned)(p.out_st * 4), stk = SUM ? (unsigned)(p.skip.st * 4) : 0u;
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const int abl = ABL ? p.ablate : 0;

  // ---- weight ring (2 slots of CHQ pair-quads = 8 CHQ records) ------------------------------------------------
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
      if (r < CH && src_vq * 8 + r < QPS * 8 && !(abl & 128)) stg[m] = bld4(rw, vlane, (unsigned)(src_rec + r) * 1024u);
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
      if (w + m * NW < CH && !(abl & 128)) *reinterpret_cast<v4f*>(lds_wr + wslot * (CH * 1024) + m * (NW * 1024)) = stg[m];
    wslot ^= 1;
  };
  auto sync = [&]() {
    if (ABL && (abl & 8))
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  issue_loads();
  stage_write();
  sync();
  issue_loads();
  const char* cb = lds_rd;
  v4f rec_const = v4f{1e-3f, -1e-3f, 2e-3f, -2e-3f};
  auto rec = [&](auto ql, int j) {
    if (ABL && (abl & 64)) return rec_const;
    return *reinterpret_cast<const v4f*>(cb + decltype(ql)::value * 8192 + j * 1024);
  };
  v4f a0 = rec(ic<0>{}, 0), a1 = rec(ic<0>{}, 1);

  v4f hold[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) hold[s] = v4f{0.f, 0.f, 0.f, 0.f};
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
#define SQUAD2(QI, B0, B1, B2, B3)                                              \
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
#define SQUAD2_1(QI, B0)                                      \
  do {                                                        \
    const v4f b0_ = rec(ic<(QI) % CHQ>{}, 4);                 \
    MFMA4(acc, a0, B0);                                       \
    MFMA4(acd, b0_, B0);                                      \
    ring_step(ic<(QI)>{});                                    \
    ring_end(ic<(QI)>{});                                     \
  } while (0)

  v4f xr[XD];
#pragma unroll
  for (int i = 0; i < XD; ++i) xr[i] = zero4;
  {
    const unsigned tt0 = rev ? p.nsteps - 1 : 0;
    static_for<XD>([&](auto v) { xr[v.value] = bld4(rx0, vo0, tt0 * st0 + 64 * v.value); });
  }

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;
    const unsigned o0 = tt * st0, o2 = tt * st2, oo = (tt + cy) * sto, ok = tt * stk;
    float xs2 = 0.f;
    if (NS2) xs2 = bld1(rx2, vo2, o2);
    if ((step > 0 || cy) && !(abl & 32)) {
      const unsigned op = (rev ? tt + 1 : tt - 1 + cy) * sto;
#pragma unroll
      for (int s = 0; s < NS; ++s) hold[s] = bld4(ro, voo, op + 64 * s);
    }

    for (int pr = 0; pr < NP; ++pr) {
      const int s0 = 2 * pr;
      v4f cprev0 = zero4, cprev1 = zero4, skip0 = zero4, skip1 = zero4;   // the second slice's are requested late (registers)
      const unsigned nx = (pr + 1 < NP ? tt : ttn) * st0;   // x of the next pair / next step
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
      static_for<NV0>([&](auto v) {
        constexpr int V = decltype(v)::value;
        const v4f xb = xr[V % XD];
        SQUAD2(1 + V, xb.x, xb.y, xb.z, xb.w);
        if (!(abl & 1)) {
          if constexpr (V + XD < NV0)
            xr[V % XD] = bld4(rx0, vo0, o0 + 64 * (V + XD));
          else
            xr[V % XD] = bld4(rx0, vo0, nx + 64 * (V + XD - NV0));   // wraps into the next pair
        }
      });
      if constexpr (NS2 > 0) SQUAD2_1(1 + NV0, xs2);
      static_for<NS>([&](auto sp) {
        constexpr int SP = decltype(sp)::value;
        SQUAD2(1 + NV0 + NS2 + SP, hold[SP].x, hold[SP].y, hold[SP].z, hold[SP].w);
        if constexpr (SP == NS - 4) {      // cell state / residual operand of the first slice: four quads ahead of their use
          if ((step > 0 || cy) && !(abl & 16)) cprev0 = bld4(rc, vlane, s0 * 1024);
          if (SUM && !(abl & 16)) skip0 = bld4(rsk, vok, ok + 64 * s0);
        }
      });
      static_for<PAD>([&](auto u) {
        ring_step(ic<QPS + decltype(u)::value>{});
        ring_end(ic<QPS + decltype(u)::value>{});
      });
      // cell updates of the two slices
      {
        const bool cheap = ABL && (abl & 2);
        const v4f ig = cheap ? acc[0] : sigmoid4<GPK>(acc[0]), fg = cheap ? acc[1] : sigmoid4<GPK>(acc[1]);
        const v4f gg = cheap ? acc[2] : tanh4<GPK>(acc[2]), og = cheap ? acc[3] : sigmoid4<GPK>(acc[3]);
        if ((step > 0 || cy) && !(abl & 16)) cprev1 = bld4(rc, vlane, s0 * 1024 + 1024);   // in flight under the first slice's gate math
        if (SUM && !(abl & 16)) skip1 = bld4(rsk, vok, ok + 64 * s0 + 64);
        const v4f cn = cheap ? fg + cprev0 + ig : cell4<GPK2>(fg, cprev0, ig, gg);
        v4f hn = cheap ? og + gg : mul_rn4<GPK2>(og, tanh4<GPK>(cn));
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));   // h + skip adds the ROUNDED h
        if (abl & 4) asm volatile("" ::"v"(cn), "v"(hn));
        if (!(abl & 4)) bst4(cn, rc, vlane, s0 * 1024);
        if (valid && !(abl & 4)) {
          bst4(hn, ro, voo, oo + 64 * s0);
          if (SUM) bst4(add_rn4<GPK2>(hn, skip0), ro2, voo2, oo + 64 * s0);
        }
      }
      {
        const bool cheap = ABL && (abl & 2);
        const v4f ig = cheap ? acd[0] : sigmoid4<GPK>(acd[0]), fg = cheap ? acd[1] : sigmoid4<GPK>(acd[1]);
        const v4f gg = cheap ? acd[2] : tanh4<GPK>(acd[2]), og = cheap ? acd[3] : sigmoid4<GPK>(acd[3]);
        const v4f cn = cheap ? fg + cprev1 + ig : cell4<GPK2>(fg, cprev1, ig, gg);
        v4f hn = cheap ? og + gg : mul_rn4<GPK2>(og, tanh4<GPK>(cn));
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));
        if (abl & 4) asm volatile("" ::"v"(cn), "v"(hn));
        if (!(abl & 4)) bst4(cn, rc, vlane, s0 * 1024 + 1024);
        if (valid && !(abl & 4)) {
          bst4(hn, ro, voo, oo + 64 * s0 + 64);
          if (SUM) bst4(add_rn4<GPK2>(hn, skip1), ro2, voo2, oo + 64 * s0 + 64);
        }
      }
    }
  }
#undef SQUAD2
#undef SQUAD2_1
}

template <int H, int NW, int M, int NV0, int NS2, int CHQ, int PAD, int MODE, int XD = 4, bool ABL = false>
int launch_static2_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = (size_t)2 * CHQ * 8192;
  static_assert(2 * CHQ * 8192 <= 160 * 1024, "ring does not fit the LDS");
  auto k = lstm_static2_kernel<H, NW, M, NV0, NS2, CHQ, PAD, MODE, XD, ABL>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_static2_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
