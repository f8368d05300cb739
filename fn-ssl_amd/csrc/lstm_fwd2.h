// Reserve-saving training forward with TWO 16-sequence groups per wave set (round 4; the narrow-band layers of the FN-SSL
// training step at config 4's shard: reference = nn.LSTM in train mode, FN-SSL/Model.py:25-29, Lightning/main.py:149-157).
//
// The forward twin of lstm_bwd2.h, for the same reason: 512 narrow-band groups on 256 CUs, four waves per group, every group
// pulling the whole weight matrix (2.1 MB at H = 256) from L2 once per step.  Here the four waves of a workgroup (one per
// SIMD) own the same quarters of the hidden slices, and every weight record they load multiplies the operands of BOTH of the
// CU's groups.  Nothing hides a memory round trip for a SIMD's only wave, so
//   * the weight quads (bias quad included) run through a 4-deep register pipeline, fully unrolled and fenced, and are the
//     only vector-memory requests of the matrix loop;
//   * the B operands come from LDS: x_t of both groups (each wave fetches a quarter of the blocks a step ahead and puts them
//     there) and h_{t-1} (every wave writes the h blocks it produces next to the global copy) — both double-buffered by step
//     parity, one workgroup barrier per step;
//   * the cell state of a wave's own slices stays in registers.
// Arithmetic: lstm_rec_kernel's — bias records as the accumulators' initial values, the input blocks, the 4-channel remainder
// quad, the recurrent blocks, in that order; the same gate code — h and the reserve are bit-identical.
#pragma once

#include "lstm_static.h"
#include "lstm_train.h"

#pragma clang fp contract(off)

namespace fnssl_lstm {

// NV0: 16-channel blocks of src0 (c0 / 16); NS2: 1 = a 4-channel src2 (block 1's raw features), 0 = none
template <int H, int NV0, int NS2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_fwd2_kernel(const LstmParams p) {
  constexpr int NS = H / 16, SPLIT = 4, NSL = NS / SPLIT, G = 2, WD = 4;
  constexpr int QPS = 1 + NV0 + NS2 + NS;                       // quads per hidden slice: bias, input, remainder, recurrent
  constexpr int NQ = NSL * QPS;                                 // quads of this wave per step
  static_assert(NQ % WD == 0, "the weight pipeline must close on itself at the step end");
  constexpr int kXBytes = G * NV0 * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [parity][x of both groups] [parity][h of both groups]
  FNSSL_GUARDED_KERNEL(p);
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g4 = lane >> 4;
  const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const unsigned st0 = (unsigned)(p.src0.st * 4), st2 = NS2 ? (unsigned)(p.src2.st * 4) : 0u, sto = (unsigned)(p.out_st * 4);

  unsigned vo0[G], vo2[G], voo[G];
  rsrc_t rx0[G], rx2[G], ro[G], rres[G], rc[G];
  bool valid[G], tvalid[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const int task = p.task0 + wg * G + k;
    tvalid[k] = task < p.task1;
    int q = task * 16 + n;
    valid[k] = q < p.nseq && tvalid[k];
    if (q >= p.nseq) q = p.nseq - 1;
    const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;
    rx0[k] = split_addr(p.src0.p, qo * p.src0.so + qi * p.src0.si, 4 * g4, vo0[k]);
    vo2[k] = 0;
    rx2[k] = NS2 ? split_addr(p.src2.p, qo * p.src2.so + qi * p.src2.si, 0, vo2[k]) : rx0[k];
    ro[k] = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g4, voo[k]);
    rres[k] = make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                        ((size_t)dir * p.ntasks + (tvalid[k] ? task : 0)) * p.nsteps * (size_t)(NS * kReserveRecs * 1024));
    rc[k] = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                      ((size_t)dir * (p.ntasks + 16) + (tvalid[k] ? task : p.ntasks + k)) * (NS * 1024));
  }

  // ---- weight pipeline: the wave's NSL slices are one contiguous run of NQ quads; ar[q % WD] holds quad q of the step
  const rsrc_t rwd = make_rsrc(p.wpack[dir]);
  // (quad indices are compile-time after unrolling; the byte offset of the wave's run is re-made opaque every step, or the
  //  compiler hoists all 4 NQ record offsets out of the step loop into scalar registers and spills 500 of them)
  unsigned wb = (unsigned)(part * NSL * p.quads_per_slice * 4) * 1024u;
  v4f ar[WD][4];
  auto dq_load = [&](v4f* dst, int quad) {   // quad: index inside the step, a constant at every call site
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = bld4(rwd, vlane, wb + (unsigned)(quad * 4 + j) * 1024u);
  };
#pragma unroll
  for (int k = 0; k < WD; ++k) dq_load(ar[k], k);

  // ---- LDS images
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f* const xs = reinterpret_cast<v4f*>(smem) + lane;                       // [parity][group][block][lane]
  v4f* const hs = reinterpret_cast<v4f*>(smem + 2 * kXBytes) + lane;         // [parity][group][slice][lane]
  auto xs_at = [&](int par, int k, int v) -> v4f& { return xs[((par * G + k) * NV0 + v) * 64]; };
  auto hs_at = [&](int par, int k, int s) -> v4f& { return hs[((par * G + k) * NS + s) * 64]; };
  // x blocks this wave fetches for the workgroup: blocks part * NV0 / 4 .. of both groups
  constexpr int XPW = NV0 / SPLIT;
  static_assert(NV0 % SPLIT == 0, "input blocks divide among the four waves");
  v4f xpre[G][XPW];
  float rem_pre[G] = {0.f, 0.f}, rem_cur[G] = {0.f, 0.f};
  auto x_issue = [&](unsigned tt) {
#pragma unroll
    for (int k = 0; k < G; ++k) {
#pragma unroll
      for (int i = 0; i < XPW; ++i) xpre[k][i] = bld4(rx0[k], vo0[k], tt * st0 + 64 * (part * XPW + i));
      if (NS2) rem_pre[k] = bld1(rx2[k], vo2[k] + 4 * g4, tt * st2);
    }
  };
  auto x_commit = [&](int par) {
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
      for (int i = 0; i < XPW; ++i) xs_at(par, k, part * XPW + i) = xpre[k][i];
  };
  // step 0: x_0 into parity 0, h_{-1} = 0 into parity 1; x_1 requested
  const unsigned tt_first = rev ? p.nsteps - 1 : 0;
  x_issue(tt_first);
#pragma unroll
  for (int k = 0; k < G; ++k)
#pragma unroll
    for (int i = 0; i < NSL; ++i) hs_at(1, k, part * NSL + i) = zero4;
  x_commit(0);
#pragma unroll
  for (int k = 0; k < G; ++k) rem_cur[k] = rem_pre[k];
  if (p.nsteps > 1) x_issue(rev ? tt_first - 1 : tt_first + 1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  v4f c_keep[G][NSL];
#pragma unroll
  for (int k = 0; k < G; ++k)
#pragma unroll
    for (int i = 0; i < NSL; ++i) c_keep[k][i] = zero4;

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const int par = step & 1;
    const unsigned oo = tt * sto;
    asm volatile("" : "+s"(wb));

    static_for<NSL>([&](auto slc) {
      constexpr int SL = decltype(slc)::value;
      const int s = part * NSL + SL;
      v4f acc[G][4];
      // one quad of the stream against one B value set per group; the slot is refilled right behind its use
      auto quad = [&](auto qc, auto&& bval) {
        constexpr int Q = decltype(qc)::value;                  // quad index inside the step
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const v4f xb = bval(k);
          MFMA4(acc[k], ar[Q % WD][0], xb.x);
          MFMA4(acc[k], ar[Q % WD][1], xb.y);
          MFMA4(acc[k], ar[Q % WD][2], xb.z);
          MFMA4(acc[k], ar[Q % WD][3], xb.w);
        }
        __builtin_amdgcn_sched_barrier(0);
        dq_load(ar[Q % WD], (Q + WD) % NQ);
        __builtin_amdgcn_sched_barrier(0);
      };
      constexpr int Q0 = SL * QPS;
      // ---- bias quad: the accumulators' initial values (the same for both groups)
#pragma unroll
      for (int k = 0; k < G; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = ar[Q0 % WD][j];
      __builtin_amdgcn_sched_barrier(0);
      dq_load(ar[Q0 % WD], (Q0 + WD) % NQ);
      __builtin_amdgcn_sched_barrier(0);
      // ---- input blocks
      static_for<NV0>([&](auto vc) {
        constexpr int V = decltype(vc)::value;
        quad(ic<Q0 + 1 + V>{}, [&](int k) { return xs_at(par, k, V); });
      });
      // ---- 4-channel remainder quad: only record 0 is real
      if constexpr (NS2) {
        constexpr int Q = Q0 + 1 + NV0;
#pragma unroll
        for (int k = 0; k < G; ++k) MFMA4(acc[k], ar[Q % WD][0], rem_cur[k]);
        __builtin_amdgcn_sched_barrier(0);
        dq_load(ar[Q % WD], (Q + WD) % NQ);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- recurrent blocks
      static_for<NS>([&](auto spc) {
        constexpr int SP = decltype(spc)::value;
        quad(ic<Q0 + 1 + NV0 + NS2 + SP>{}, [&](int k) { return hs_at(par ^ 1, k, SP); });
      });
      // ---- cell update (PyTorch gate order i, f, g, o), reserve, h
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const v4f ig = sigmoid4(acc[k][0]);
        const v4f fg = sigmoid4(acc[k][1]);
        const v4f gg = tanh4(acc[k][2]);
        const v4f og = sigmoid4(acc[k][3]);
        const v4f cn = cell4(fg, c_keep[k][SL], ig, gg);
        v4f hn = mul_rn4(og, tanh4(cn));
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));
        c_keep[k][SL] = cn;
        if (tvalid[k]) {
          const unsigned rb = (tt * NS + s) * (kReserveRecs * 1024);
          bst4(ig, rres[k], vlane, rb);
          bst4(fg, rres[k], vlane, rb + 1024);
          bst4(gg, rres[k], vlane, rb + 2048);
          bst4(og, rres[k], vlane, rb + 3072);
          bst4(cn, rres[k], vlane, rb + 4096);
        }
        hs_at(par, k, s) = hn;
        if (valid[k]) bst4(hn, ro[k], voo[k], oo + 64 * s);
      }
    });

    // ---- x_{t+1} into the other parity, x_{t+2} requested; then everybody meets
    if (step + 1 < p.nsteps) {
      x_commit(par ^ 1);
#pragma unroll
      for (int k = 0; k < G; ++k) rem_cur[k] = rem_pre[k];
      if (step + 2 < p.nsteps) x_issue(rev ? tt - 2 : tt + 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // the final cell state where the other kernels leave it (one record per group and slice)
#pragma unroll
  for (int k = 0; k < G; ++k)
#pragma unroll
    for (int i = 0; i < NSL; ++i) bst4(c_keep[k][i], rc[k], vlane, (part * NSL + i) * 1024);
}

template <int H, int NV0, int NS2>
int launch_fwd2_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;
  const size_t lds = (size_t)2 * 2 * NV0 * 1024 + (size_t)2 * 2 * (H / 16) * 1024;
  auto k = lstm_fwd2_kernel<H, NV0, NS2>;
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_fwd2_kernel");
  return FNSSL_OK;
}

extern template int launch_fwd2_k<256, 16, 0>(const LstmParams&, int, hipStream_t);
extern template int launch_fwd2_k<256, 16, 1>(const LstmParams&, int, hipStream_t);

}  // namespace fnssl_lstm
