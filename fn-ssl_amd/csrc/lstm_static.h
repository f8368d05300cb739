// Shape-specialised twin of lstm_rec_kernel (lstm_kernel.h) for the layer shapes of the
// FN-SSL network itself.  Same formulation, same packed stream, same k-ordered fp32 MFMA
// chain per sequence (results are bit-identical to the generic kernel), but the whole
// structure of a hidden slice is known at compile time:
//   * the x-part / h-part loops are fully unrolled, so every ring read is
//     `chunk base VGPR + immediate` and the per-quad scalar bookkeeping and branches of the
//     generic ring reader disappear (micro-benchmark tools/ubench/mfma_peak.hip: a lone
//     wave issues MFMAs at 99 % of the pipe rate without them, 89 % with them);
//   * ring commits sit at fixed positions (chunk ends coincide with slice ends);
//   * the x operand ring is 4 blocks deep with a fixed 3-quad prefetch distance.
// Inputs are restricted to what the fused forward needs: one summed input tensor of C0
// channels (C0 = 16*NV0 or the 4-channel remainder NS0 = 1), an optional 4-channel
// concatenated tensor (NS2 = 1) and the optional fused residual output.
#pragma once

#include "lstm_kernel.h"

#pragma clang fp contract(off)

namespace fnssl_lstm {

template <int I>
using ic = std::integral_constant<int, I>;

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<N, I + 1>(f);
  }
}

// ABL = true: timing-ablation twin driven by FNSSL_ABLATE (bits as in lstm_rec_kernel); wrong results.
template <int H, int NW, int M, int NV0, int NS0, int NS2, int CHQ, int PAD, int MODE, bool ABL = false, int XD = 4,
          int NV2 = 0>
__global__ void __launch_bounds__(NW * 64) lstm_static_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  constexpr int NS = H / 16;
  constexpr bool HAS2 = (MODE & kHas2) != 0, SUM = (MODE & kSum) != 0;
  static_assert(!(MODE & kHas1), "static kernel: single summed input only");
  static_assert(HAS2 == (NS2 + NV2 > 0), "NS2 / NV2 must match the mode");
  constexpr int QPS = 1 + NV0 + NS0 + NV2 + NS2 + NS;  // real quads per slice
  constexpr int VQ = QPS + PAD;                         // virtual quads per slice
  static_assert(VQ % CHQ == 0, "chunks must tile the (padded) slice");
  constexpr int CH = 4 * CHQ;                           // records per chunk
  static_assert(CH <= NW * M, "chunk does not fit the staging registers");
  static_assert(NV0 == 0 || NV0 % XD == 0, "the x ring depth must divide the block count");
  static_assert(!(NS0 && NV0), "remainder-only or block-only summed input");
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
  const unsigned vo2v = vo2 + 12 * g;   // 16-channel blocks of the concatenated input: lane (n, g) reads 4g..4g+3
  const rsrc_t rsk = SUM ? split_addr(p.skip.p, qo * p.skip.so + qi * p.skip.si, dir * H + 4 * g, vok) : rx0;
  const rsrc_t ro = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo);
  const rsrc_t ro2 = SUM ? split_addr(p.out_sum, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo2) : ro;
  const rsrc_t rc = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                              ((size_t)dir * (p.ntasks + 16) + (task < p.task1 ? task : p.ntasks + w)) * (NS * 1024));
  const unsigned cy = (unsigned)p.carry;   // streaming (see lstm_rec_kernel)
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned st0 = (unsigned)(p.src0.st * 4), st2 = HAS2 ? (unsigned)(p.src2.st * 4) : 0u;
  const unsigned sto = (unsigned)(p.out_st * 4), stk = SUM ? (unsigned)(p.skip.st * 4) : 0u;
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const int abl = ABL ? p.ablate : 0;
  if (NS0) vo0 -= 12 * g;   // remainder-only input: lane (n, g) reads channel g, not 4g..4g+3

  // ---- weight ring ---------------------------------------------------------------
  char* const lds_rd = smem + lane * 16;
  char* const lds_wr = smem + w * 1024 + lane * 16;
  constexpr int NSLOT = 2;
  int wslot = 0;          // slot the staged chunk is committed to
  int rslot = 0;          // slot being read
  int src_rec = 0;        // real record index of the next chunk to stage
  int src_vq = 0;         // its virtual quad offset inside the slice
  v4f stg[M];
  auto issue_loads = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int r = w + m * NW;
      if (r < CH && src_vq * 4 + r < QPS * 4) stg[m] = bld4(rw, vlane, (unsigned)(src_rec + r) * 1024u);
    }
    src_vq += CHQ;
    src_rec += CH;
    if (src_vq == VQ) {
      src_vq = 0;
      src_rec -= PAD * 4;                       // the padding quads do not exist in the stream
      if (src_rec == NS * QPS * 4) src_rec = 0;
    }
  };
  // The staged chunk goes to the slot that was read during the PREVIOUS chunk period, which is
  // free for the whole current period, so the wait-for-data + ds_write sits in the middle of the
  // chunk (where waves are not lined up at a barrier and the in-order vmcnt wait costs least)
  // and the chunk end is only `lgkmcnt(0); s_barrier`.
  auto stage_write = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (w + m * NW < CH) *reinterpret_cast<v4f*>(lds_wr + wslot * (CH * 1024) + m * (NW * 1024)) = stg[m];
    wslot = (wslot + 1 == NSLOT) ? 0 : wslot + 1;
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
  const char* cb = lds_rd;                      // base of the chunk being read
  auto rec = [&](auto ql, int j) {              // record j of quad ql (compile time) of the current chunk
    return *reinterpret_cast<const v4f*>(cb + decltype(ql)::value * 4096 + j * 1024);
  };
  v4f a0 = rec(ic<0>{}, 0), a1 = rec(ic<0>{}, 1);

  v4f hold[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) hold[s] = v4f{0.f, 0.f, 0.f, 0.f};
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f acc[4];

  // end-of-quad ring step for the quad with virtual index QI inside the slice: peek the next
  // quad's first two records (same chunk) or, at a chunk end, commit and re-read
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
      rslot = (rslot + 1 == NSLOT) ? 0 : rslot + 1;
      cb = lds_rd + rslot * (CH * 1024);
      a0 = rec(ic<0>{}, 0);
      a1 = rec(ic<0>{}, 1);
    }
  };
#define SQUAD(QI, B0, B1, B2, B3)                                               \
  do {                                                                          \
    const v4f a2_ = rec(ic<(QI) % CHQ>{}, 2), a3_ = rec(ic<(QI) % CHQ>{}, 3);   \
    __builtin_amdgcn_sched_barrier(0);                                          \
    MFMA4(acc, a0, B0);                                                         \
    MFMA4(acc, a1, B1);                                                         \
    ring_step(ic<(QI)>{});                                                      \
    __builtin_amdgcn_sched_barrier(0);                                          \
    MFMA4(acc, a2_, B2);                                                        \
    MFMA4(acc, a3_, B3);                                                        \
    ring_end(ic<(QI)>{});                                                       \
  } while (0)
#define SQUAD1(QI, B0)         \
  do {                         \
    MFMA4(acc, a0, B0);        \
    ring_step(ic<(QI)>{});     \
    ring_end(ic<(QI)>{});      \
  } while (0)

  // ---- x operand ring: block v lives in xr[v % XD], requested XD-1 quads before its use --
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
    const unsigned o0 = tt * st0, o2 = tt * st2, oo = (tt + cy) * sto, ok = tt * stk;
    float xs0[NS0 > 0 ? NS0 : 1], xs2[NS2 > 0 ? NS2 : 1];   // 4-channel remainder blocks: lane (n, g) holds channel 4u + g
    static_for<NS0>([&](auto u) { xs0[u.value] = bld1(rx0, vo0, o0 + 64 * NV0 + 16 * u.value); });
    static_for<NS2>([&](auto u) { xs2[u.value] = bld1(rx2, vo2, o2 + 64 * NV2 + 16 * u.value); });
    v4f xv2[NV2 > 0 ? NV2 : 1];   // the concatenated input is the same for every slice: held for the whole step
    static_for<NV2>([&](auto v) { xv2[v.value] = bld4(rx2, vo2v, o2 + 64 * v.value); });
    if ((step > 0 || cy) && !(abl & 32)) {
      const unsigned op = (rev ? tt + 1 : tt - 1 + cy) * sto;
#pragma unroll
      for (int s = 0; s < NS; ++s) hold[s] = bld4(ro, voo, op + 64 * s);
    }

    for (int s = 0; s < NS; ++s) {
      v4f cprev = zero4, skipv = zero4;
      const unsigned nx = (s + 1 < NS ? tt : ttn) * st0;   // x of the next slice / next step

      // quad 0: bias -> accumulators
      acc[0] = a0;
      acc[1] = a1;
      acc[2] = rec(ic<0>{}, 2);
      acc[3] = rec(ic<0>{}, 3);
      ring_step(ic<0>{});
      ring_end(ic<0>{});
      // summed input, 16 channels per quad
      static_for<NV0>([&](auto v) {
        constexpr int V = decltype(v)::value;
        const v4f xb = xr[V % XD];
        SQUAD(1 + V, xb.x, xb.y, xb.z, xb.w);
        if (!(abl & 1)) {
          if constexpr (V + XD < NV0)
            xr[V % XD] = bld4(rx0, vo0, o0 + 64 * (V + XD));
          else
            xr[V % XD] = bld4(rx0, vo0, nx + 64 * (V + XD - NV0));   // wraps into the next slice
        }
      });
      // cell state and residual operand of this slice: requested after the x part (whose ring
      // registers they reuse), still a whole recurrent part ahead of their use
      if ((step > 0 || cy) && !(abl & 16)) cprev = bld4(rc, vlane, s * 1024);
      if (SUM && !(abl & 16)) skipv = bld4(rsk, vok, ok + 64 * s);
      static_for<NS0>([&](auto u) { SQUAD1(1 + NV0 + decltype(u)::value, xs0[decltype(u)::value]); });
      static_for<NV2>([&](auto v) {
        constexpr int V = decltype(v)::value;
        SQUAD(1 + NV0 + NS0 + V, xv2[V].x, xv2[V].y, xv2[V].z, xv2[V].w);
      });
      static_for<NS2>([&](auto u) { SQUAD1(1 + NV0 + NS0 + NV2 + decltype(u)::value, xs2[decltype(u)::value]); });
      // recurrent part
      static_for<NS>([&](auto sp) {
        constexpr int SP = decltype(sp)::value;
        SQUAD(1 + NV0 + NS0 + NV2 + NS2 + SP, hold[SP].x, hold[SP].y, hold[SP].z, hold[SP].w);
      });
      // ring padding
      static_for<PAD>([&](auto u) {
        ring_step(ic<QPS + decltype(u)::value>{});
        ring_end(ic<QPS + decltype(u)::value>{});
      });
      // cell update
      v4f cn, hn;
      if (ABL && (abl & 2)) {
        cn = acc[1] + cprev + acc[0];
        hn = acc[3] + acc[2];
      } else {
        const v4f ig = sigmoid4(acc[0]);
        const v4f fg = sigmoid4(acc[1]);
        const v4f gg = tanh4(acc[2]);
        const v4f og = sigmoid4(acc[3]);
        cn = cell4(fg, cprev, ig, gg);
        hn = mul_rn4(og, tanh4(cn));
      }
      asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));   // h + skip adds the ROUNDED h
      if (!(abl & 4)) {
        bst4(cn, rc, vlane, s * 1024);
        if (valid) {
          bst4(hn, ro, voo, oo + 64 * s);
          if (SUM) bst4(add_rn4(hn, skipv), ro2, voo2, oo + 64 * s);
        }
      } else {
        asm volatile("" ::"v"(cn), "v"(hn));
      }
    }
  }
#undef SQUAD
#undef SQUAD1
}

template <int H, int NW, int M, int NV0, int NS0, int NS2, int CHQ, int PAD, int MODE, bool ABL = false, int XD = 4,
          int NV2 = 0>
int launch_static_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = (size_t)2 * CHQ * 4096;
  static_assert(2 * CHQ * 4096 <= 160 * 1024, "ring does not fit the LDS");
  auto k = lstm_static_kernel<H, NW, M, NV0, NS0, NS2, CHQ, PAD, MODE, ABL, XD, NV2>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_static_kernel");
  return FNSSL_OK;
}

// Return kNoStatic when no specialisation exists for (c0, c2, mode, NW).
constexpr int kNoStatic = -100;
// Returned by the cluster-kernel launchers when the device cannot hold every member workgroup at once (occupancy
// query): the caller runs the per-wave / pair-split kernels instead, unguarded.
constexpr int kNoCluster = -101;

// Knobs of the cluster kernels' bounded waits (fnssl_tuning): CLUSTER_SPIN_LIMIT (spins before a wave gives up),
// CLUSTER_TEST_STALL = m + 1 (fault injection: member m of cluster 0 exits at once, as if it never became resident).
inline unsigned cluster_spin_limit() {
  const int v = fnssl::tune(FNSSL_TUNE_CLUSTER_SPIN_LIMIT, 1, 1 << 30);
  return v ? (unsigned)v : (1u << 20);
}
// compute units the cluster kernels may count on: the device's minus what the caller keeps busy elsewhere (RESERVED_CUS:
// RCCL's all-reduce kernels under an overlapped backward), in whole XCD-uniform steps (a multiple of 8 CUs)
inline int cluster_cus() {
  const int ncu = fnssl::device_cus();
  int r = fnssl::tune(FNSSL_TUNE_RESERVED_CUS, 1, ncu);
  r = (r + 7) / 8 * 8;
  return r >= ncu ? 0 : ncu - r;
}
// blocks the device can hold at once for this kernel (occupancy query x CUs) >= grid?  One query per call: cheap (host only).
inline bool cluster_grid_fits(const void* kernel, int threads, size_t lds, int grid) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return (long long)per_cu * cluster_cus() >= grid;
}
inline int cluster_test_stall() { return fnssl::tune(FNSSL_TUNE_CLUSTER_TEST_STALL, 1, 1 << 20) - 1; }
int launch_static_h128(const LstmParams& p, int mode, int NW, int nwg, hipStream_t st);
int launch_static_h256(const LstmParams& p, int mode, int NW, int nwg, hipStream_t st);
int launch_static3_h256(const LstmParams& p, int mode, int nwg, hipStream_t st);   // lstm_static3.h, pair-interleaved stream
int launch_static4_h256(const LstmParams& p, int mode, int nwg, hipStream_t st);   // lstm_static4.h, quad-interleaved stream
int launch_static_ipdnet(const LstmParams& p, int mode, int H, int NW, int nwg, hipStream_t st);

}  // namespace fnssl_lstm
