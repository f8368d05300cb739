// Shape-specialised kernel for launches with FEW 16-sequence groups (the training step's per-GPU shard):
// the compile-time slice structure of lstm_static.h (immediate LDS offsets, fixed ring commit points,
// 4-deep operand ring) combined with the hidden-slice split of lstm_rec_kernel — SPLIT waves of a
// workgroup share one group, wave part p computes slices [p*NS/SPLIT, (p+1)*NS/SPLIT) of every step,
// the ring carries super-quads (the same quad position of the SPLIT slices in flight), and all waves
// re-read the complete h_{t-1} from the output tensor after a workgroup barrier at the step end.
// Inputs: one tensor of 16*NV0 (+ NS0 = 1: a 4-channel one instead) channels, optionally a concatenated
// 4-channel tensor (NS2 = 1); MODE may carry kSave (training: gates + cell state to the reserve).
// Same k-ordered fp32 MFMA chain per sequence as every other fp32 kernel: bit-identical results.
#pragma once

#include "lstm_static.h"

namespace fnssl_lstm {

// DIRECT = true: no LDS ring — each wave streams its own quads of the weight stream straight from L2 through a
// 4-quad-deep register pipeline (launches with few waves leave the L2 bandwidth to spare, and the ring's
// workgroup barriers — one per CHQ quads — are what limits the split geometries); CHQ / PAD / M are unused.
// NV2 = 16-channel blocks of the concatenated input (IPDnet), held in registers for the whole step.
template <int H, int NW, int M, int SPLIT, int NV0, int NS0, int NS2, int CHQ, int PAD, int MODE, bool DIRECT = false,
          int NV2 = 0>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && !DIRECT ? 3 : 1)) lstm_split_static_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  constexpr int NS = H / 16, NSL = NS / SPLIT;
  constexpr bool HAS2 = (MODE & kHas2) != 0, SAVE = (MODE & kSave) != 0, SUM = (MODE & kSum) != 0;
  static_assert(!(MODE & kHas1), "single summed input");
  static_assert(HAS2 == (NS2 + NV2 > 0) && NS2 <= 1 && NS0 <= 1 && !(NS0 && NV0), "input segments");
  static_assert(NS % SPLIT == 0 && NW % SPLIT == 0, "split geometry");
  constexpr int QPS = 1 + NV0 + NS0 + NV2 + NS2 + NS;  // real quads per slice
  constexpr int VQ = QPS + (DIRECT ? 0 : PAD);          // virtual quads per slice
  static_assert(DIRECT || VQ % CHQ == 0, "chunks must tile the (padded) slice");
  static_assert(!DIRECT || (NSL * QPS) % 4 == 0, "DIRECT: the 4-deep register pipeline must close once per step");
  constexpr int CH = 4 * CHQ * SPLIT;                   // records per chunk (CHQ super-quads)
  constexpr int QB = 4096 * SPLIT;                      // bytes per super-quad in the ring
  static_assert(CH <= NW * M, "chunk does not fit the staging registers");
  constexpr int XD = 4;
  static_assert(NV0 == 0 || NV0 % XD == 0, "the x ring depth must divide the block count");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int part = w % SPLIT;
  const int task = p.task0 + wg * (NW / SPLIT) + w / SPLIT;
  const bool tvalid = task < p.task1;
  int q = task * 16 + n;
  const bool valid = q < p.nseq && tvalid;
  if (q >= p.nseq) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  unsigned vo0 = 0, vo2 = 0, voo = 0, vok = 0, voo2 = 0;
  const rsrc_t rx0 = split_addr(p.src0.p, qo * p.src0.so + qi * p.src0.si, 4 * g, vo0);
  const rsrc_t rx2 = HAS2 ? split_addr(p.src2.p, qo * p.src2.so + qi * p.src2.si, g, vo2) : rx0;
  const unsigned vo2v = vo2 + 12 * g;   // 16-channel blocks of the concatenated input: lane (n, g) reads 4g..4g+3
  const rsrc_t ro = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo);
  const rsrc_t rsk = SUM ? split_addr(p.skip.p, qo * p.skip.so + qi * p.skip.si, dir * H + 4 * g, vok) : rx0;
  const rsrc_t ro2 = SUM ? split_addr(p.out_sum, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo2) : ro;
  const unsigned stk = SUM ? (unsigned)(p.skip.st * 4) : 0u;
  const rsrc_t rc = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                              ((size_t)dir * (p.ntasks + 16) + (tvalid ? task : p.ntasks + w)) * (NS * 1024));
  const rsrc_t rres = SAVE ? make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                                       ((size_t)dir * p.ntasks + (tvalid ? task : 0)) * p.nsteps *
                                           (size_t)(NS * kReserveRecs * 1024))
                           : rc;
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned st0 = (unsigned)(p.src0.st * 4), st2 = HAS2 ? (unsigned)(p.src2.st * 4) : 0u;
  const unsigned sto = (unsigned)(p.out_st * 4);
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  if (NS0) vo0 -= 12 * g;   // remainder-only input: lane (n, g) reads channel g, not 4g..4g+3

  // ---- weight ring of super-quads ------------------------------------------------------
  char* const lds_rd = smem + lane * 16 + part * 4096;
  char* const lds_wr = smem + w * 1024 + lane * 16;
  int wslot = 0, rslot = 0;
  int src_slice = 0;      // local slice index of the next chunk to stage
  int src_vq = 0;         // its first virtual quad inside the slice
  v4f stg[M];
  auto issue_loads = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int r = w + m * NW;                      // record inside the chunk
      const int sq = src_vq + r / (4 * SPLIT);       // quad position inside the slice
      const int pr = (r / 4) % SPLIT;                // which part's quad
      if (r < CH && sq < QPS)
        stg[m] = bld4(rw, vlane, (unsigned)(((pr * NSL + src_slice) * QPS + sq) * 4 + (r & 3)) * 1024u);
    }
    src_vq += CHQ;
    if (src_vq == VQ) {
      src_vq = 0;
      src_slice = src_slice + 1 == NSL ? 0 : src_slice + 1;
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (w + m * NW < CH) *reinterpret_cast<v4f*>(lds_wr + wslot * (CH * 1024) + m * (NW * 1024)) = stg[m];
    wslot ^= 1;
  };
  auto sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const char* cb = lds_rd;
  auto rec = [&](auto ql, int j) {
    return *reinterpret_cast<const v4f*>(cb + decltype(ql)::value * QB + j * 1024);
  };
  v4f a0, a1;
  if constexpr (!DIRECT) {
    issue_loads();
    stage_write();
    sync();
    issue_loads();
    a0 = rec(ic<0>{}, 0);
    a1 = rec(ic<0>{}, 1);
  }
  // DIRECT: ar[k] holds quad (cursor + k) of this wave's cyclic share of the stream: local slices
  // [part*NSL, (part+1)*NSL), QPS quads each, contiguous in the standard stream
  v4f ar[4][4];
  unsigned dq_off = 0;                                       // byte offset of the next quad to request
  const unsigned dq_base = (unsigned)(part * NSL * QPS) * 4096u, dq_len = (unsigned)(NSL * QPS) * 4096u;
  auto dq_load = [&](auto k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ar[decltype(k)::value][j] = bld4(rw, vlane, dq_base + dq_off + j * 1024u);
    dq_off = dq_off + 4096u == dq_len ? 0u : dq_off + 4096u;
  };
  if constexpr (DIRECT) {
    dq_load(ic<0>{});
    dq_load(ic<1>{});
    dq_load(ic<2>{});
    dq_load(ic<3>{});
  }

  v4f hold[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) hold[s] = v4f{0.f, 0.f, 0.f, 0.f};
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f acc[4];

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
  // (SL = the local slice being unrolled; in DIRECT mode quad SL*QPS + QI of the step lives in ar[.. % 4])
#define SQUAD(QI, B0, B1, B2, B3)                                                 \
  do {                                                                            \
    if constexpr (DIRECT) {                                                       \
      constexpr int K_ = (SL * QPS + (QI)) % 4;                                   \
      MFMA4(acc, ar[K_][0], B0);                                                  \
      MFMA4(acc, ar[K_][1], B1);                                                  \
      MFMA4(acc, ar[K_][2], B2);                                                  \
      MFMA4(acc, ar[K_][3], B3);                                                  \
      dq_load(ic<K_>{});                                                          \
    } else {                                                                      \
      const v4f a2_ = rec(ic<(QI) % CHQ>{}, 2), a3_ = rec(ic<(QI) % CHQ>{}, 3);   \
      __builtin_amdgcn_sched_barrier(0);                                          \
      MFMA4(acc, a0, B0);                                                         \
      MFMA4(acc, a1, B1);                                                         \
      ring_step(ic<(QI)>{});                                                      \
      __builtin_amdgcn_sched_barrier(0);                                          \
      MFMA4(acc, a2_, B2);                                                        \
      MFMA4(acc, a3_, B3);                                                        \
      ring_end(ic<(QI)>{});                                                       \
    }                                                                             \
  } while (0)
#define SQUAD1(QI, B0)                          \
  do {                                          \
    if constexpr (DIRECT) {                     \
      constexpr int K_ = (SL * QPS + (QI)) % 4; \
      MFMA4(acc, ar[K_][0], B0);                \
      dq_load(ic<K_>{});                        \
    } else {                                    \
      MFMA4(acc, a0, B0);                       \
      ring_step(ic<(QI)>{});                    \
      ring_end(ic<(QI)>{});                     \
    }                                           \
  } while (0)

  v4f xr[XD];
#pragma unroll
  for (int i = 0; i < XD; ++i) xr[i] = zero4;
  {
    const unsigned tt0 = rev ? p.nsteps - 1 : 0;
    static_for<(NV0 < XD ? NV0 : XD)>([&](auto v) { xr[v.value] = bld4(rx0, vo0, tt0 * st0 + 64 * v.value); });
  }

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;
    const unsigned o0 = tt * st0, o2 = tt * st2, oo = tt * sto, ok = tt * stk;
    float xs0 = 0.f, xs2 = 0.f;
    if (NS0) xs0 = bld1(rx0, vo0, o0);
    if (NS2) xs2 = bld1(rx2, vo2, o2 + 64 * NV2);
    v4f xv2[NV2 > 0 ? NV2 : 1];   // the concatenated input is the same for every slice: held for the whole step
    static_for<NV2>([&](auto v) { xv2[v.value] = bld4(rx2, vo2v, o2 + 64 * v.value); });
    if (step > 0) {
      const unsigned op = (rev ? tt + 1 : tt - 1) * sto;
#pragma unroll
      for (int s = 0; s < NS; ++s) hold[s] = bld4(ro, voo, op + 64 * s);
    }

    static_for<NSL>([&](auto slc) {
      constexpr int SL = decltype(slc)::value;
      const unsigned sg = (unsigned)(part * NSL + SL);          // global hidden slice of this wave
      v4f cprev = zero4;
      const unsigned nx = (SL + 1 < NSL ? tt : ttn) * st0;      // x of the next slice / next step
      if constexpr (DIRECT) {   // bias quad
        constexpr int K0 = (SL * QPS) % 4;
        acc[0] = ar[K0][0];
        acc[1] = ar[K0][1];
        acc[2] = ar[K0][2];
        acc[3] = ar[K0][3];
        dq_load(ic<K0>{});
      } else {
        acc[0] = a0;
        acc[1] = a1;
        acc[2] = rec(ic<0>{}, 2);
        acc[3] = rec(ic<0>{}, 3);
        ring_step(ic<0>{});
        ring_end(ic<0>{});
      }
      static_for<NV0>([&](auto v) {
        constexpr int V = decltype(v)::value;
        const v4f xb = xr[V % XD];
        SQUAD(1 + V, xb.x, xb.y, xb.z, xb.w);
        if constexpr (V + XD < NV0)
          xr[V % XD] = bld4(rx0, vo0, o0 + 64 * (V + XD));
        else
          xr[V % XD] = bld4(rx0, vo0, nx + 64 * (V + XD - NV0));   // wraps into the next slice
      });
      if (step > 0) cprev = bld4(rc, vlane, sg * 1024);
      v4f skipv = zero4;
      if (SUM) skipv = bld4(rsk, vok, ok + 64 * sg);
      if constexpr (NS0 > 0) SQUAD1(1 + NV0, xs0);
      static_for<NV2>([&](auto v) {
        constexpr int V2 = decltype(v)::value;
        SQUAD(1 + NV0 + NS0 + V2, xv2[V2].x, xv2[V2].y, xv2[V2].z, xv2[V2].w);
      });
      if constexpr (NS2 > 0) SQUAD1(1 + NV0 + NS0 + NV2, xs2);
      static_for<NS>([&](auto sp) {
        constexpr int SP = decltype(sp)::value;
        SQUAD(1 + NV0 + NS0 + NV2 + NS2 + SP, hold[SP].x, hold[SP].y, hold[SP].z, hold[SP].w);
      });
      if constexpr (!DIRECT)
        static_for<PAD>([&](auto u) {
          ring_step(ic<QPS + decltype(u)::value>{});
          ring_end(ic<QPS + decltype(u)::value>{});
        });
      const v4f ig = sigmoid4(acc[0]);
      const v4f fg = sigmoid4(acc[1]);
      const v4f gg = tanh4(acc[2]);
      const v4f og = sigmoid4(acc[3]);
      const v4f cn = cell4(fg, cprev, ig, gg);
      v4f hn = mul_rn4(og, tanh4(cn));
      asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));   // h + skip adds the ROUNDED h
      bst4(cn, rc, vlane, sg * 1024);
      if (valid) {
        bst4(hn, ro, voo, oo + 64 * sg);
        if (SUM) bst4(add_rn4(hn, skipv), ro2, voo2, oo + 64 * sg);
      }
      if (SAVE && tvalid) {
        const unsigned rb = (tt * NS + sg) * (kReserveRecs * 1024);
        bst4(ig, rres, vlane, rb);
        bst4(fg, rres, vlane, rb + 1024);
        bst4(gg, rres, vlane, rb + 2048);
        bst4(og, rres, vlane, rb + 3072);
        bst4(cn, rres, vlane, rb + 4096);
      }
    });
    // the partner waves read my h slices at the start of the next step: stores performed, then meet
    if constexpr (SPLIT > 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
#undef SQUAD
#undef SQUAD1
}

template <int H, int NW, int M, int SPLIT, int NV0, int NS0, int NS2, int CHQ, int PAD, int MODE, bool DIRECT = false,
          int NV2 = 0>
int launch_split_static_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = DIRECT ? 0 : (size_t)2 * CHQ * SPLIT * 4096;
  static_assert(2 * CHQ * SPLIT * 4096 <= 160 * 1024, "ring does not fit the LDS");
  auto k = lstm_split_static_kernel<H, NW, M, SPLIT, NV0, NS0, NS2, CHQ, PAD, MODE, DIRECT, NV2>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_split_static_kernel");
  return FNSSL_OK;
}

// kNoStatic when (H, nw, split, c0, c2, mode, chunk cap) has no instantiation; max_chq = LDS budget in super-quads
int launch_split_static(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st);

}  // namespace fnssl_lstm
