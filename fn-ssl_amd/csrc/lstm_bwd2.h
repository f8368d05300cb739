// Back-propagation through time with TWO 16-sequence groups per wave set (round 4; the narrow-band layers of the FN-SSL
// training step at config 4's shard: reference = autograd through nn.LSTM, FN-SSL/Lightning/main.py:149-157).
//
// Why: config 4's shard has 512 narrow-band groups on 256 CUs.  lstm_bwd_kernel<256, 8, 1, 4, DIRECT> gives a group four
// waves, each streaming its quarter of [W_ih | W_hh]^T (2 MB at c0g = H = 256) from L2 once per step for its 16 sequences:
// 1 GB per step over the chip, 12.6 TB/s of L2 demand at the measured 74 ms per step — the launch sits at 0.65 of the fp32 MFMA
// roof whatever the phases of the co-resident groups do (profiles/r04).  Here the four waves of a workgroup own the
// same quarters but every weight record they load multiplies the dA rows of BOTH of the CU's groups: half the L2 traffic per
// flop.  A SIMD then holds ONE wave, so nothing hides a memory round trip for it — which is why
//   * the weight quads run through a 4-deep register pipeline (8-deep measured the same) and are the ONLY
//     vector-memory requests of the matrix loop: the dA rows both groups need as B operands go through LDS (128 KiB, written
//     by phase A next to the global copy the weight-gradient kernel reads), and
//   * the read-only operands of phase A (forward reserve, upstream gradient) of the NEXT step are requested at the start of
//     the current step's matrix phase for the first group (lstm_bwdc.h's lesson: requested when needed they were the longest
//     item of the step), the second group's at the top of phase A behind the first group's arithmetic;
//   * the carried dh goes through LDS as well (32 KiB: 160 KiB in all), the carried dc and c_t stay in registers, and the two
//     barriers of a step wait for LDS only: no vector-memory drain anywhere in the step loop.
// Arithmetic: lstm_bwd_kernel's phase-A expressions and its k order per output block — bit-identical dA and dx.
#pragma once

#include "lstm_static.h"
#include "lstm_train.h"

#pragma clang fp contract(off)

namespace fnssl_lstm {

template <int H>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_bwd2_kernel(const BwdParams p) {
  constexpr int NS = H / 16, SPLIT = 4, NSL = NS / SPLIT, NVB = 4 * H / 16, G = 2, WD = 4;
  constexpr int NPRE = NSL;             // (group, slice) operand sets requested a step ahead: group 0's; group 1's at the top of phase A, behind
                                        // group 0's arithmetic (all of them ahead, or half of group 1's too: 2 - 4 % slower)
  constexpr int NLATE = G * NSL - NPRE > 0 ? G * NSL - NPRE : 1;
  // LDS: the dA rows of both groups, [group][block][lane] float4 (128 KiB), then the carried dh of both groups,
  // [group][hidden slice][lane] (32 KiB) — 160 KiB, the whole CU's
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kXsBytes = G * NVB * 1024;
  if (p.guard && __hip_atomic_load(p.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // uniform for the grid
  if (p.guard && p.fallback_count && blockIdx.x == 0 && threadIdx.x == 0 && p.task0 == 0) atomicAdd(p.fallback_count, 1u);
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g4 = lane >> 4;
  const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const unsigned sdh = (unsigned)(p.dh.st * 4), sda = (unsigned)(p.da_st * 4), sdx = (unsigned)(p.dx_st * 4);
  const int nsol = (p.co_pad >> 6) / SPLIT;   // output slices (4 x 16 channels each) of this wave
  const int hq = p.co_pad >> 2;               // channels per output quarter

  // ---- the two groups of this workgroup
  unsigned vdh[G], vda[G], vdx[G];
  rsrc_t rdh[G], rda[G], rdx[G], rres[G], rsc[G];
  bool valid[G], tvalid[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const int task = p.task0 + wg * G + k;
    tvalid[k] = task < p.task1;
    int q = task * 16 + n;
    valid[k] = q < p.nseq && tvalid[k];
    if (q >= p.nseq) q = p.nseq - 1;
    const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;
    rdh[k] = split_addr(p.dh.p, qo * p.dh.so + qi * p.dh.si, dir * H + 4 * g4, vdh[k]);
    rda[k] = split_addr(p.da, qo * p.da_so + qi * p.da_si, dir * 4 * H + 4 * g4, vda[k]);
    vdx[k] = 0;
    rdx[k] = p.c0g ? split_addr(p.dx, qo * p.dx_so + qi * p.dx_si, dir * p.c0g + 4 * g4, vdx[k]) : rdh[k];
    rres[k] = make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                        ((size_t)dir * p.ntasks + (tvalid[k] ? task : 0)) * p.nsteps * (size_t)(NS * kReserveRecs * 1024));
    rsc[k] = make_rsrc(reinterpret_cast<const char*>(p.scratch) +
                       ((size_t)dir * (p.ntasks + 16) + (tvalid[k] ? task : p.ntasks + k)) * (2 * NS * 1024));
  }

  // ---- weight pipeline: this wave's data quads (the zero "bias" quad of each output slice is skipped) form a cyclic sequence
  // of nsol * NVB quads; ar[k] holds quad (cursor + k) of it
  const rsrc_t rwd = make_rsrc(p.wpack[dir]);
  const int dq_total = nsol * NVB;
  int dq_next = 0;
  v4f ar[WD][4];
  auto dq_load = [&](v4f* dst) {
    const int sl = dq_next / NVB, qq = dq_next - sl * NVB;
    const unsigned rec0 = (unsigned)(((part * nsol + sl) * p.quads_per_slice + 1 + qq) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = bld4(rwd, vlane, (rec0 + j) * 1024u);
    dq_next = dq_next + 1 == dq_total ? 0 : dq_next + 1;
  };
#pragma unroll
  for (int k = 0; k < WD; ++k) dq_load(ar[k]);

  // ---- phase-A operands of one (group, hidden slice): forward reserve + upstream gradient
  struct Ops {
    v4f ig, fg, gg, og, cp, dhu;   // (c_t itself is the c_{t-1} of the step processed before: kept in registers)
  };
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  const v4f one = v4f{1.f, 1.f, 1.f, 1.f};
  auto issue = [&](int k, int step, int s) {
    const unsigned tt = rev ? step : p.nsteps - 1 - step;
    const bool has_prev = step + 1 < p.nsteps;
    const unsigned tp = has_prev ? (rev ? tt + 1 : tt - 1) : tt;
    const unsigned rb = (tt * NS + s) * (kReserveRecs * 1024);
    Ops o;
    o.ig = bld4(rres[k], vlane, rb);
    o.fg = bld4(rres[k], vlane, rb + 1024);
    o.gg = bld4(rres[k], vlane, rb + 2048);
    o.og = bld4(rres[k], vlane, rb + 3072);
    o.dhu = bld4(rdh[k], vdh[k], tt * sdh + 64 * s);
    o.cp = zero4;
    if (has_prev) o.cp = bld4(rres[k], vlane, (tp * NS + s) * (kReserveRecs * 1024) + 4096);
    return o;
  };
  Ops pre[NPRE];
  v4f ct_keep[G][NSL], dc_keep[G][NSL];   // c_t of the step about to be processed; the carried dc (this wave's own slices)
#pragma unroll
  for (int k = 0; k < G; ++k)
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      if (k * NSL + sl < NPRE) pre[k * NSL + sl] = issue(k, 0, part * NSL + sl);
      const unsigned tt0 = rev ? 0 : p.nsteps - 1;
      ct_keep[k][sl] = bld4(rres[k], vlane, (tt0 * NS + part * NSL + sl) * (kReserveRecs * 1024) + 4096);
      dc_keep[k][sl] = zero4;
    }
  v4f* const dhs = reinterpret_cast<v4f*>(smem + kXsBytes) + lane;   // [group][slice][lane]

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? step : p.nsteps - 1 - step;
    const unsigned oa = tt * sda;

    // ---- phase A: gate gradients of my hidden slices, both groups
    Ops late[NLATE];
#pragma unroll
    for (int i = NPRE; i < G * NSL; ++i) late[(i - NPRE) % NLATE] = issue(i / NSL, step, part * NSL + i % NSL);
#pragma unroll
    for (int k = 0; k < G; ++k) {
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        const int s = part * NSL + sl;
        const Ops& in = k * NSL + sl < NPRE ? pre[(k * NSL + sl) % NPRE] : late[(k * NSL + sl + NLATE * NPRE - NPRE) % NLATE];
        const v4f ig = in.ig, fg = in.fg, gg = in.gg, og = in.og, ct = ct_keep[k][sl], cp = in.cp;
        v4f dhc = zero4;
        if (step > 0) dhc = dhs[(k * NS + s) * 64];   // the previous step's matrix phase left it there (barrier since)
        v4f dh = in.dhu + dhc, dc = dc_keep[k][sl];
        const v4f tc = tanh4(ct);
        dc += dh * og * (one - tc * tc);
        const v4f dao = dh * tc * og * (one - og);
        const v4f dai = dc * gg * ig * (one - ig);
        const v4f daf = dc * cp * fg * (one - fg);
        const v4f dag = dc * ig * (one - gg * gg);
        dc_keep[k][sl] = dc * fg;
        ct_keep[k][sl] = cp;
        {   // the matrix phase takes its B operands from LDS: block (gate, slice) of the row, this lane's 16 bytes
          v4f* const xs = reinterpret_cast<v4f*>(smem) + (size_t)k * NVB * 64 + lane;
          xs[(0 * NS + s) * 64] = dai;
          xs[(1 * NS + s) * 64] = daf;
          xs[(2 * NS + s) * 64] = dag;
          xs[(3 * NS + s) * 64] = dao;
        }
        if (valid[k]) {
          bst4(dai, rda[k], vda[k], oa + 64 * s);
          bst4(daf, rda[k], vda[k], oa + 4 * H + 64 * s);
          bst4(dag, rda[k], vda[k], oa + 8 * H + 64 * s);
          bst4(dao, rda[k], vda[k], oa + 12 * H + 64 * s);
        }
      }
    }
    // the B operands below are the dA blocks the four waves of the workgroup have just put into LDS
    // (LDS only: nothing of this loop waits for vector memory except the uses of what it loaded — the weight pipeline and
    //  the operand prefetch run through the steps undrained)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- next step's read-only operands travel during the matrix phase
    if (step + 1 < p.nsteps) {
#pragma unroll
      for (int i = 0; i < NPRE; ++i) pre[i] = issue(i / NSL, step + 1, part * NSL + i % NSL);
    }

    // ---- phase B: [dx | dh_prev]^T = [W_ih | W_hh]^T da^T for both groups against one stream of weight records
    const v4f* const xs = reinterpret_cast<const v4f*>(smem) + lane;
    v4f xc[G], xn[G];
#pragma unroll
    for (int k = 0; k < G; ++k) xc[k] = xs[(size_t)k * NVB * 64];
    for (int sol = 0; sol < nsol; ++sol) {
      const int so = part * nsol + sol;
      v4f acc[G][4];
#pragma unroll
      for (int k = 0; k < G; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = zero4;
      // fully unrolled and fenced: left to itself the compiler gathers the loads of a loop body at its end and waits for ALL
      // of them at the top of the next trip — the pipeline then covers nothing (lstm_bwd_kernel's DIRECT loop has exactly
      // that s_waitcnt vmcnt(0) per four quads).  The only vector-memory requests of the loop are the weight records.
      static_for<NVB>([&](auto vc) {
        constexpr int V = decltype(vc)::value;
#pragma unroll
        for (int k = 0; k < G; ++k) xn[k] = xs[((size_t)k * NVB + ((V + 1) & (NVB - 1))) * 64];   // (wraps into the next output slice)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < G; ++k) {
          MFMA4(acc[k], ar[V % WD][0], xc[k].x);
          MFMA4(acc[k], ar[V % WD][1], xc[k].y);
          MFMA4(acc[k], ar[V % WD][2], xc[k].z);
          MFMA4(acc[k], ar[V % WD][3], xc[k].w);
        }
        __builtin_amdgcn_sched_barrier(0);
        dq_load(ar[V % WD]);
#pragma unroll
        for (int k = 0; k < G; ++k) xc[k] = xn[k];
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int k = 0; k < G; ++k)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int ob = qq * hq + 16 * so;                 // first channel of this 16-channel block
          if (ob < p.c0g) {
            if (valid[k]) bst4(acc[k][qq], rdx[k], vdx[k], tt * sdx + 4 * ob);
          } else if (ob < p.c0g + H) {
            dhs[(k * NS + ((ob - p.c0g) >> 4)) * 64] = acc[k][qq];
          }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // carried dh visible to the workgroup
  }
}

template <int H>
int launch_bwd2_k(const BwdParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;
  const size_t lds = (size_t)2 * (4 * H / 16) * 1024 + (size_t)2 * (H / 16) * 1024;   // both groups' dA rows + carried dh
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd2_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(lstm_bwd2_kernel<H>, dim3(nwg), dim3(256), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_bwd2_kernel");
  return FNSSL_OK;
}

extern template int launch_bwd2_k<256>(const BwdParams&, int, hipStream_t);

}  // namespace fnssl_lstm
