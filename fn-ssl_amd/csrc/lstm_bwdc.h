// Cluster-resident back-propagation through time for the H = 128 full-band layers of the FN-SSL training step (round 4;
// BASELINE config 4: reference = autograd through nn.LSTM, FN-SSL/Lightning/main.py:149-157).
//
// Why: lstm_bwd_kernel gives the waves of a workgroup fixed 16-sequence groups, so a launch is paced by its fullest SIMD —
// config 4's shard has 1200 full-band groups, 2 waves each = 2.34 waves per SIMD, run as 3: the layer sat at 0.44 of the fp32
// MFMA roof.  The forward of the same layers left that pigeonhole through lstm_f32c.h; this is the same move for the
// backward pass.  Per step and group the backward is  [dx_t | dh_{t-1}]^T = [W_ih | W_hh]^T da_t^T  — a product whose OUTPUT
// range (c0g + H = 384 channels = 6 "output slices" of 4 x 16 channels in the packed stream's fake-LSTM arrangement,
// lstm_train.h) splits over a cluster of 6 CUs: member m keeps output slice m of the transposed weight stream (32 quads =
// 128 KiB) in LDS for the whole launch, and the cluster's groups are work items its 12 waves take in turn; the groups beyond
// a multiple of the wave count change hands every step (a group's state — carried dh, dc — lives in memory), so every SIMD
// carries the same load over the launch.  A group-step of a member:
//   phase A (VALU)  the gate gradients of the hidden slices whose dh_{t-1} THIS member produces in phase B (so the carried
//                   state never crosses CUs): slice 2 + m for every member, plus slice 0 / 1 for members 4 / 5 — written to
//                   the dA tensor (sc1 write-through stores), then the member's tag of (group) is set to step + 1.  The
//                   read-only operands (forward reserve, upstream gradient) were requested one group-step ahead;
//   phase B (MFMA)  once all six tags show the step: the whole dA row of the group (32 blocks through an 8-deep operand
//                   ring; plain loads — the row is read once per CU, after it is complete) against the member's weight
//                   slice; dx_t blocks go to the input-gradient tensor, dh_{t-1} blocks to the group's carried-state
//                   record.  One wave per SIMD at a time (an LDS token): the matrix pipe is shared round-robin whatever the
//                   priorities, and waves that became ready together otherwise run their matrix phases together and then
//                   wait on memory together.
// Same stream, same k order per output block, same phase-A expressions as lstm_bwd_kernel: bit-identical dA and dx.
// Waits are bounded and cooperative (status word, no trap), the call enqueues lstm_bwd_kernel behind this kernel as its
// guarded fallback — exactly as fnssl_lstm_forward does for lstm_f32c.h (include/fnssl.h).
#pragma once

#include "lstm_static.h"
#include "lstm_train.h"

#pragma clang fp contract(off)

namespace fnssl_lstm {

constexpr int kBwdcWaves = 12;   // waves per member; with the 8-deep dA ring (144 registers) 3 % faster than 16 waves and a 4-deep ring

struct BwdClusterParams {
  unsigned* tags;       // [cluster][group in cluster][16]: words 0..7 = per-member "dA of step s written" (s + 1), 8..15 = "phase B of
                        // step s done" (s + 1); zeroed before the launch
  unsigned* status;     // one word, zeroed before the launch: 0 = fine, else the code of the first wave that gave up
  int clusters_per_dir;
  int groups_per_cluster;
  int members;          // CUs per cluster = output slices of the layer (co_pad / 64)
  int clusters_per_xcd; // 32 / members
  unsigned spin_limit;
  int stall_member;     // test knob: this member of cluster 0 exits at once (-1: none)
  int rotate;
  int simd_token;       // one wave per SIMD in the matrix phase at a time (FNSSL_BWDC_NO_TOKEN=1: off)
  int no_prefetch;      // A/B knob (FNSSL_BWDC_NO_PREFETCH): phase-A operands requested when needed
  int ablate;           // ablate build only (FNSSL_BWDC_ABLATE): timing experiments, wrong results by construction
};

// ABLRT (make ABLATE=1 only): run-time ablation bits cp.ablate — 1 no tag waits, 2 no phase-A loads, 4 no phase-A stores,
// 8 no dA loads in phase B, 16 no output stores, 32 no drain in front of the "phase B done" tag, 64 no phase A at all,
// 128 L1-bypassing dA loads, 256 plain dA stores, 512 per-phase cycle counters (printed for cluster 0, waves 0 and 5)
template <int NW_ = kBwdcWaves, bool ABLRT = false, int XD_ = 8>
__global__ void __launch_bounds__(NW_ * 64) lstm_bwdc_kernel(const BwdParams p, const BwdClusterParams cp) {
  constexpr int H = 128, NS = H / 16, NW = NW_;
  const int ABL = ABLRT ? cp.ablate : 0;
  constexpr int NVB = 4 * H / 16;                             // 16-channel blocks of a dA row = data quads per output slice
  constexpr int XD = XD_;                                    // depth of the dA operand ring (16-channel blocks in flight)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int b = blockIdx.x;
  const int kx = b >> 3;                                      // index inside the XCD (blocks are dealt round-robin: speed only)
  const int m = kx % cp.members;                              // member = output slice
  const int cx = kx / cp.members;
  if (cx >= cp.clusters_per_xcd) return;                      // CUs that do not make a whole cluster on their XCD
  const int cl = cx * 8 + (b & 7);
  const int ncl = cp.clusters_per_dir * p.ndir;
  if (cl >= ncl) return;
  const int dir = cl / cp.clusters_per_dir;
  const int ck = cl - dir * cp.clusters_per_dir;
  const int g0 = ck * cp.groups_per_cluster;
  const int g1 = g0 + cp.groups_per_cluster < p.ntasks ? g0 + cp.groups_per_cluster : p.ntasks;
  if (__hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  if (cl == 0 && m == cp.stall_member) return;                // test knob: a member that never shows up

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool rev = dir == 1;
  const unsigned vlane = lane * 16;

  {   // my output slice of the transposed weight stream -> LDS, once (the slice's leading "bias" quad is all zeros: skipped)
    const int qps = 1 + NVB;
    const v4f* src = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(p.wpack[dir]) + ((size_t)m * qps + 1) * 4096);
    v4f* dst = reinterpret_cast<v4f*>(smem);
    for (int i = threadIdx.x; i < NVB * 4 * 64; i += NW * 64) dst[i] = src[i];
  }
  // one "matrix phase" token per SIMD (wave w runs on SIMD w & 3), behind the weight slice
  unsigned* const token = reinterpret_cast<unsigned*>(smem + NVB * 4096) + (w & 3);
  if (threadIdx.x < 4) reinterpret_cast<unsigned*>(smem + NVB * 4096)[threadIdx.x] = 0;
  __syncthreads();
  const char* const lds_rd = smem + lane * 16;
  auto rec = [&](int q, int j) { return *reinterpret_cast<const v4f*>(lds_rd + (q * 4 + j) * 1024); };

  // which hidden slices this member owns in phase A = the ones whose dh_{t-1} its phase B produces: output block qq of
  // output slice m covers channels qq hq + 16 m .. + 15 of [dx | dh]
  const int hq = p.co_pad >> 2;
  int own[4], nown = 0;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int ob = qq * hq + 16 * m;
    own[qq] = (ob >= p.c0g && ob < p.c0g + H) ? (ob - p.c0g) >> 4 : -1;
    nown += own[qq] >= 0;
  }
  (void)nown;
  // the (up to) two slices whose read-only operands are requested one group-step ahead
  int q0 = -1, q1 = -1;
#pragma unroll
  for (int qq = 3; qq >= 0; --qq) {
    if (own[qq] >= 0) {
      q1 = q0;
      q0 = qq;
    }
  }

  const unsigned sdh = (unsigned)(p.dh.st * 4), sda = (unsigned)(p.da_st * 4), sdx = (unsigned)(p.dx_st * 4);
  unsigned* const tag_cl = cp.tags + (size_t)cl * cp.groups_per_cluster * 16;
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  const v4f one4 = v4f{1.f, 1.f, 1.f, 1.f};

  if (g0 + w >= g1) return;                                   // (a wave without a group: nobody waits for it)
  bool dead = false;
  auto bounded_wait = [&](auto ready, auto reload, unsigned code) {
    for (unsigned spins = 0; !ready(); ++spins) {
      __builtin_amdgcn_s_sleep(8);
      reload();
      const bool out = spins > cp.spin_limit;
      if (out || (spins & 63) == 63) {
        if (out) {
          if (lane == 0) __hip_atomic_store(cp.status, code | (unsigned)(cl & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return false;
        }
        if (__hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      }
    }
    return true;
  };

  // schedule of a wave within a step: its regular group g0 + w + i NW (i < FULL / NW), then at most one leftover group
  // g0 + FULL + j, j = (w - step) mod NW < LEFT (lstm_f32c.h)
  const int NGR = g1 - g0;
  const bool rot = cp.rotate && NGR >= NW;
  const int FULL = rot ? (NGR / NW) * NW : NGR, LEFT = NGR - FULL;

  // per-lane addressing of a group (lstm_f32c.h "locate"): the wave's first sequence is its lowest address
  struct Grp {
    unsigned vdh, vda, vdx;
    bool valid;
  };
  auto locate = [&](int task, Grp& gr, rsrc_t& rdh, rsrc_t& rda, rsrc_t& rdx) {
    const int q0 = task * 16;
    const int qo0 = q0 / p.q_inner, qi0 = q0 - qo0 * p.q_inner;
    gr.valid = q0 + n < p.nseq;
    const int d = q0 + n < p.nseq ? n : p.nseq - 1 - q0;
    const bool crossed = qi0 + d >= p.q_inner;
    auto one = [&](const float* base, long long so, long long si, int extra, unsigned& voff) {
      const long long delta = (long long)d * si + (crossed ? so - (long long)p.q_inner * si : 0ll);
      voff = (unsigned)(delta * 4) + (unsigned)(extra * 4);
      return make_rsrc(base + ((long long)qo0 * so + (long long)qi0 * si));
    };
    rdh = one(p.dh.p, p.dh.so, p.dh.si, dir * H + 4 * g, gr.vdh);
    rda = one(p.da, p.da_so, p.da_si, dir * 4 * H + 4 * g, gr.vda);
    gr.vdx = 0;
    rdx = p.c0g ? one(p.dx, p.dx_so, p.dx_si, dir * p.c0g + 4 * g, gr.vdx) : rdh;
  };

  int task = g0 + w, step = 0;
  unsigned* pub_tag = nullptr;       // deferred "phase B done" tag of the previous group-step
  unsigned pub_val = 0;
  auto pub_flush = [&]() {
    if (pub_tag) {
      if (!(ABL & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(pub_tag, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pub_tag = nullptr;
    }
  };

  // read-only operands of one (group, step, hidden slice) of phase A: forward reserve + upstream gradient
  struct Ops {
    v4f ig, fg, gg, og, ct, cp, dhu;
  };
  auto issue = [&](int task_, int step_, rsrc_t rdh_, unsigned vdh_, int s_) {
    const unsigned tt_ = rev ? step_ : p.nsteps - 1 - step_;
    const bool has_prev_ = step_ + 1 < p.nsteps;
    const unsigned tp_ = has_prev_ ? (rev ? tt_ + 1 : tt_ - 1) : tt_;
    const rsrc_t rres_ = make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                                   ((size_t)dir * p.ntasks + task_) * p.nsteps * (size_t)(NS * kReserveRecs * 1024));
    const unsigned rb = (tt_ * NS + s_) * (kReserveRecs * 1024);
    Ops o;
    o.ig = bld4(rres_, vlane, rb);
    o.fg = bld4(rres_, vlane, rb + 1024);
    o.gg = bld4(rres_, vlane, rb + 2048);
    o.og = bld4(rres_, vlane, rb + 3072);
    o.ct = bld4(rres_, vlane, rb + 4096);
    o.dhu = bld4(rdh_, vdh_, tt_ * sdh + 64 * s_);
    o.cp = zero4;
    if (has_prev_) o.cp = bld4(rres_, vlane, (tp_ * NS + s_) * (kReserveRecs * 1024) + 4096);
    return o;
  };
  // requested a group-step ahead (behind the tag store of the current one, in front of its tag wait): they travel while the
  // wave waits for the other members and runs its matrix phase — the two HBM round trips of a two-slice member were the
  // longest item of the cluster's lock step (profiles/r04/i_bwdc_phase_times*)
  Ops pre0, pre1;
  bool have_pre = false;
  auto advance = [&](int task_, int step_, int& task_n, int& step_n) {
    task_n = task_;
    step_n = step_;
    if (task_ < g0 + FULL && task_ + NW < g0 + FULL) {
      task_n = task_ + NW;
    } else {
      int j = (w - step_) % NW;
      if (j < 0) j += NW;
      if (task_ < g0 + FULL && j < LEFT) {
        task_n = g0 + FULL + j;
      } else {
        task_n = g0 + w;
        step_n = step_ + 1;
      }
    }
  };

  // ablate build, bit 512: where a wave's time goes (shader-clock cycles per phase, summed over its group-steps)
  const bool TIMED = ABLRT && (ABL & 512);
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0, t_begin = 0;
  unsigned nitems = 0;
  auto lap = [&](int k) {
    if (TIMED) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[k] += now - tlast;
      tlast = now;
    }
  };
  if (TIMED) t_begin = tlast = __builtin_amdgcn_s_memtime();
  for (;;) {
    Grp gr;
    rsrc_t rdh, rda, rdx;
    locate(task, gr, rdh, rda, rdx);
    const unsigned tt = rev ? step : p.nsteps - 1 - step;      // the forward direction's gradient flows T-1 .. 0
    const unsigned oa = tt * sda;
    const rsrc_t rsc = make_rsrc(reinterpret_cast<const char*>(p.scratch) +
                                 ((size_t)dir * (p.ntasks + 16) + task) * (2 * NS * 1024));
    unsigned* const tag_g = tag_cl + (size_t)(task - g0) * 16;
    pub_flush();
    // a leftover group's previous step ran on ANOTHER wave of this member: its carried state is final once that wave's
    // "phase B done" tag shows the step (a regular group is this wave's own: program order)
    if (task >= g0 + FULL && step > 0 && !dead && !(ABL & 1)) {
      unsigned tv = __hip_atomic_load(tag_g + 8 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dead = !bounded_wait([&]() { return tv >= (unsigned)step; },
                           [&]() { tv = __hip_atomic_load(tag_g + 8 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, 0x60000u);
    }
    lap(0);   // drain of the previous item's output stores + hand-over wait
    // A wave that gave up on a hand-off (or saw the launch draining) stops HERE: it stores and publishes nothing computed from
    // operands it did not wait for (the guarded fallback launch of the same call rewrites every output)
    if (dead) break;

    // ---- phase A: gate gradients of my hidden slices (the expressions of lstm_bwd_kernel, in its order) -------------
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int s = own[qq];
      if (s < 0 || (ABL & 64)) continue;
      v4f ig = one4 * 0.5f, fg = ig, gg = ig, og = ig, ct = ig, dhu = ig;
      v4f cprev = zero4, dhc = zero4, dcc = zero4;
      if (!(ABL & 2)) {
        Ops o;
        if (have_pre && qq == q0) {
          o = pre0;
        } else if (have_pre && qq == q1) {
          o = pre1;
        } else {
          o = issue(task, step, rdh, gr.vdh, s);
        }
        ig = o.ig, fg = o.fg, gg = o.gg, og = o.og, ct = o.ct, dhu = o.dhu, cprev = o.cp;
        if (step > 0) {   // the carried state: written by the previous step's matrix phase, never requested ahead
          dhc = bld4_l2(rsc, vlane, s * 1024);
          dcc = bld4_l2(rsc, vlane, (NS + s) * 1024);
        }
      }
      if (TIMED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(1);   // phase-A operands arriving
      }
      v4f dh = dhu + dhc, dc = dcc;
      const v4f tc = tanh4(ct);
      dc += dh * og * (one4 - tc * tc);
      const v4f dao = dh * tc * og * (one4 - og);
      const v4f dai = dc * gg * ig * (one4 - ig);
      const v4f daf = dc * cprev * fg * (one4 - fg);
      const v4f dag = dc * ig * (one4 - gg * gg);
      if (!(ABL & 4)) bst4(dc * fg, rsc, vlane, (NS + s) * 1024);
      if (gr.valid && !(ABL & 4)) {   // write-through: the other members read these rows in their phase B
        if (ABL & 256) {
          bst4(dai, rda, gr.vda, oa + 64 * s);
          bst4(daf, rda, gr.vda, oa + 4 * H + 64 * s);
          bst4(dag, rda, gr.vda, oa + 8 * H + 64 * s);
          bst4(dao, rda, gr.vda, oa + 12 * H + 64 * s);
        } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, dai), rda, gr.vda, oa + 64 * s, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, daf), rda, gr.vda, oa + 4 * H + 64 * s, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, dag), rda, gr.vda, oa + 8 * H + 64 * s, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, dao), rda, gr.vda, oa + 12 * H + 64 * s, 16);
        }
      }
    }
    lap(2);   // gate arithmetic + issuing the stores
    // publish my dA rows of (group, step): payload stores -> vmcnt(0) -> tag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lap(3);   // store drain
    if (lane == 0) __hip_atomic_store(tag_g + m, (unsigned)step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int task_n, step_n;
    advance(task, step, task_n, step_n);
    have_pre = step_n < p.nsteps && !(ABL & 2) && !cp.no_prefetch;
    auto prefetch_next = [&]() {   // called once the last dA block of this group-step has been requested (in-order returns:
      if (have_pre) {              //  requested earlier, these HBM reads would hold up the ring's L2 hits behind them)
        Grp gn;
        rsrc_t rdhn, rdan, rdxn;
        locate(task_n, gn, rdhn, rdan, rdxn);
        pre0 = issue(task_n, step_n, rdhn, gn.vdh, own[q0 & 3]);
        if (q1 >= 0) pre1 = issue(task_n, step_n, rdhn, gn.vdh, own[q1 & 3]);
      }
    };

    // ---- phase B: my output slice of [dx | dh_prev]^T = [W_ih | W_hh]^T da^T -------------------------------------
    if (!dead && !(ABL & 1)) {
      unsigned tv = __hip_atomic_load(tag_g + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool mine = (lane & 7) < cp.members;
      dead = !bounded_wait([&]() { return __builtin_amdgcn_ballot_w64(mine && tv < (unsigned)step + 1) == 0; },
                           [&]() { tv = __hip_atomic_load(tag_g + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, 0x50000u);
    }
    lap(4);   // waiting for the other members' tags
    if (dead) break;   // (as above: no output stores, no "phase B done" tag from rows that were not complete)
    __builtin_amdgcn_s_setprio(2);   // matrix phase above the gate arithmetic of the SIMD's other waves (none / reversed: +4 %; by wave rank: no effect)
    v4f xr[XD];
    // plain loads: a dA row is read by this CU once, after all of it has been announced — no line of it can sit in this CU's
    // L1 from before (stores do not allocate), and the second 64-byte block of a line then comes from L1 (22.4 -> 21.3 ms)
    static_for<XD>([&](auto v) {
      xr[v.value] = (ABL & 8) ? one4 : (ABL & 128) ? bld4_l2(rda, gr.vda, oa + 64 * v.value) : bld4(rda, gr.vda, oa + 64 * v.value);
    });
    // One wave of a SIMD at a time in the matrix phase.  The matrix pipe is shared round-robin whatever the wave
    // priorities, so the three (four) waves of a SIMD that became ready together ran their matrix phases together — each
    // three times as long — and then stood in their memory round trips together, the pipe idle (profiles/r04/
    // i_bwdc_phase_times*: matrix phase 45 k cycles for 16.4 k of MFMAs, 30 k cycles of waits per group-step on top).
    // With the token a wave's matrix phase runs alone at the pipe's rate and the others' waits hide behind it.
    bool owned = false;    // a wave that runs out of its wait (or sees the launch aborting) goes on WITHOUT the token and must
                           // not release one another wave holds
    if (cp.simd_token) {   // (two waves per SIMD at a time, as a counting semaphore: 19.0 against 18.1 ms)
      for (unsigned spins = 0; spins < cp.spin_limit; ++spins) {
        const unsigned old = __hip_atomic_exchange(token, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // all lanes: one word
        if (__builtin_amdgcn_ballot_w64(old == 0u) != 0) {   // some lane saw it free: the wave owns it now
          owned = true;
          break;
        }
        // the launch is draining (another wave gave up on a hand-off): do not sit out the whole limit
        if ((spins & 63u) == 63u && __hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    lap(6);   // waiting for the SIMD's token
    v4f acc[4] = {zero4, zero4, zero4, zero4};
    v4f ra[4];
    static_for<4>([&](auto j) { ra[j.value] = rec(0, j.value); });
    static_for<NVB>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      const v4f xb = xr[V % XD];
      MFMA4(acc, ra[0], xb.x);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (V + 1 < NVB) ra[0] = rec(V + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(acc, ra[1], xb.y);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (V + 1 < NVB) ra[1] = rec(V + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(acc, ra[2], xb.z);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (V + 1 < NVB) ra[2] = rec(V + 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(acc, ra[3], xb.w);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (V + 1 < NVB) ra[3] = rec(V + 1, 3);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (V + XD < NVB)
        if (!(ABL & 8)) xr[V % XD] = (ABL & 128) ? bld4_l2(rda, gr.vda, oa + 64 * (V + XD)) : bld4(rda, gr.vda, oa + 64 * (V + XD));
      if constexpr (V + XD == NVB - 1) prefetch_next();   // (behind the matrix phase instead: 19.5 against 18.1 ms)
    });
    __builtin_amdgcn_s_setprio(0);
    if (owned) __hip_atomic_store(token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int ob = qq * hq + 16 * m;                          // first channel of this 16-channel block
      if (ABL & 16) {
        if (acc[qq].x == 123.456f) bst4(acc[qq], rsc, vlane, 0);
      } else if (ob < p.c0g) {
        if (gr.valid) bst4(acc[qq], rdx, gr.vdx, tt * sdx + 4 * ob);
      } else if (ob < p.c0g + H) {
        bst4(acc[qq], rsc, vlane, ((ob - p.c0g) >> 4) * 1024);
      }
    }
    lap(5);   // matrix phase (dA loads, MFMAs), output stores issued
    ++nitems;
    pub_val = (unsigned)step + 1;
    pub_tag = tag_g + 8 + m;
    if (dead) break;

    // ---- next work item
    task = task_n;
    step = step_n;
    if (step >= p.nsteps) break;
  }
  pub_flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (TIMED && cl == 0 && (w == 0 || w == 5) && lane == 0) {
    const unsigned long long total = __builtin_amdgcn_s_memtime() - t_begin;
    printf("bwdc member %d wave %d: %u group-steps, cycles per group-step: total %llu = prev-store drain + hand-over %llu, operand wait %llu, "
           "gates + store issue %llu, store drain %llu, tag wait %llu, token wait %llu, matrix phase %llu\n",
           m, w, nitems, total / nitems, tacc[0] / nitems, tacc[1] / nitems, tacc[2] / nitems, tacc[3] / nitems, tacc[4] / nitems,
           tacc[6] / nitems, tacc[5] / nitems);
  }
}

template <int NW_ = kBwdcWaves, bool ABLRT = false, int XD_ = 8>
int launch_bwdc_k(const BwdParams& p, const BwdClusterParams& cp, hipStream_t st) {
  const size_t lds = (size_t)(4 * 128 / 16) * 4096 + 64;        // 32 quads = 128 KiB, + the SIMD tokens
  auto k = lstm_bwdc_kernel<NW_, ABLRT, XD_>;
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nwg = 8 * cp.clusters_per_xcd * cp.members;   // block b: XCD b & 7, index b >> 3 inside it
  if (!cluster_grid_fits(reinterpret_cast<const void*>(k), NW_ * 64, lds, nwg)) return kNoCluster;
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_backward_plan: the family is reported only after the occupancy check, like the forward's
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW_ * 64), lds, st, p, cp);
  FNSSL_CHECK_LAUNCH("lstm_bwdc_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
