// Cluster-resident fp32 LSTM recurrence for the H = 128 full-band layers of FN-SSL (BASELINE config 2, the headline).
//
// Why: lstm_static_kernel gives every WAVE a 16-sequence group for the whole recurrence, so a launch is paced by its
// fullest SIMD: config 2's full-band layers are 7200 groups on 1024 SIMDs = 7.03 per SIMD, run as rounds of 15 + 14 waves
// per CU = 8 wave-times (pigeonhole: 0.879 of the roof before any other loss; measured 0.78).  Here the hidden SLICES of
// the layer are split over a cluster of 8 CUs — member m keeps slice m of the packed weight stream (25 quads = 100 KiB,
// or 10 quads for block 1) in LDS for the whole launch — and the 16-sequence groups become work items that the member's
// 16 waves take in turn: 225 groups per cluster = 14 per wave and one left over: the fullest SIMD runs 57 group-steps per
// step against 56.25 on average = 0.987 of the roof.  Every member computes
// its 16 hidden units for ALL groups of the cluster; h_t travels between the members through the OUTPUT tensor itself
// (the row of step t is written once and read at step t + 1 by everybody, exactly where lstm_static_kernel re-reads its
// own h): write-through (sc1) stores, sc1 loads, and one tag word per (group, member) = step + 1.  A hand-off has a
// whole step (14 group-steps, ~0.3 ms) to arrive.  A second tag check keeps every member within two group-steps of the
// others, so that seven of the eight find a group's input rows in the XCD's L2 instead of fetching them from memory
// again.  Waits are bounded (status word + trap), as in lstm_bf16c.h.  Measured: 88.6 ms per layer against 48.0 + 46.5 for
// the two rounds (profiles/r03/o_*), config 2 548 against 563 ms per step.
//
// Arithmetic: per group and slice the MFMA chain is lstm_static_kernel's — accumulators start from the bias quad, input
// quads, recurrent quads, k-ordered — and the gate math is the same code: results are bit-identical.
//
// Round 5: the same kernel for H = 256 (HH: the uni-directional narrow-band layers of the online model — 16 hidden slices
// over clusters of 16 CUs, 33 / 34 quads = 132 / 136 KiB of LDS per member, the concatenated 4 data channels of block 1 as
// a remainder quad read from src2) and for FEW groups (f32c_handles: one utterance = 6 groups per cluster): with a weight
// slice resident in LDS a step costs the member's own matrix work plus one hand-off, where the several-waves-per-group
// kernels stream the whole 1 - 2 MB matrix from L2 per group and step (one 4-mic utterance: 21 -> ~4 ms per narrow-band
// layer, tools/latency_bench.py).
//
// Gate split (GS = 4), for launches of a FEW groups per cluster (one 2-mic utterance, a 12-frame streaming chunk: one group
// per cluster, i.e. one wave per CU): the four waves of a "slot" — one per SIMD — share a group and each runs the MFMA chain
// of ONE gate (component `gate` of every weight record: the very chain MFMA4 runs for that gate, so the bits are the same),
// applies that gate's activation, and hands the activated gate to its three siblings through LDS (double-buffered 1-KiB
// records, one sequence counter per slot); all four then form c_t and h_t redundantly and the gate-0 wave stores and
// publishes.  A group-step's matrix phase shrinks from 4 x 16 x QPS MFMAs on one SIMD to 16 x QPS on each of four.
#pragma once

#include "lstm_static.h"

namespace fnssl_lstm {

constexpr int kF32cWaves = 16;   // four per SIMD (measured: 100.8 against 102.3 ms per layer with 12 = three per SIMD); 225 groups = 14 each + 1

struct F32ClusterParams {
  unsigned* tags;       // [direction][cluster][group in cluster][member H / 16], zeroed before the launch
  unsigned* status;     // one word, zeroed before the launch: 0 = fine, else the code of the first wave that gave up
  int clusters_per_dir;
  int groups_per_cluster;
  unsigned spin_limit;  // spins (~1.5 us each) a wave waits for a tag before it gives up (cluster_spin_limit())
  int stall_member;     // test knob: this member of cluster 0 exits at once (-1: none)
  int rotate;           // 1: the groups a cluster has beyond a multiple of its waves change hands every step (see the schedule)
  int prio_mode;        // 2 (default): s_setprio 2 during the matrix phase, 0 during the cell update — a wave in its matrix
                        // phase outranks its SIMD neighbours' cell updates at issue; 0: none (FNSSL_F32C_PRIO=9).  Measured
                        // (profiles/r04): 256-channel layers 88.9 -> 86.5 ms, block 1's layer 36.1 -> 35.3 ms; the reverse
                        // order, a static rank per wave and a staggered start were measured as well: equal or worse.
};

// HH: hidden size (128: clusters of 8; 256: clusters of 16 — NW_ = 16 with h_{t-1} streamed through the operand ring (STRH below),
// NW_ <= 8 with the row held, 64 registers: the gate-split form and launches of up to 8 groups per cluster);
// NV0: 16-channel blocks of the summed input; NS0: one 4-channel remainder quad — of src0 (block 1's full-band layer, NV0
// = 0) or, with MODE & kHas2, of the concatenated src2 behind the NV0 blocks of src0 (block 1's narrow-band layer);
// MODE: kSum / kHas2 / kSave bits
// ABLRT = true: timing-ablation twin (make ABLATE=1 only, wrong results) driven by the bits of FNSSL_F32C_ABL at run time:
//   1 one group's addressing for all, 2 cheap gates, 4 no tag waits, 8 no input loads, 16 no recurrent-operand loads,
//   32 no stores, 64 no cell-state / residual loads, 128 no LDS record reads in the quads, 256 no tag loads / publishes
template <int HH, int NV0, int NS0, int MODE, bool ABLRT = false, int DRIFT = 2, int NW_ = kF32cWaves, int GS = 1>
__global__ void __launch_bounds__(NW_ * 64) lstm_f32c_kernel(const LstmParams p, const F32ClusterParams cp) {
  constexpr int H = HH, NS = H / 16, NW = NW_;
  constexpr int NWS = NW / GS;                                           // schedule lanes: waves, or 4-wave slots (gate split)
  static_assert((GS == 1 || GS == 4) && NW % GS == 0 && (NWS & (NWS - 1)) == 0, "gate split: whole slots, a power of two of them");
  constexpr bool SUM = (MODE & kSum) != 0, SAVE = (MODE & kSave) != 0;   // SAVE: training forward (gates + cell state -> reserve)
  constexpr bool CAT = (MODE & kHas2) != 0;                              // the remainder quad comes from src2
  constexpr int QPS = 1 + NV0 + NS0 + NS;
  // DRIFT: a wave starts the recurrent part of its k-th group of a step only when EVERY member has finished its
  // (k - DRIFT)-th: the eight members then read a group's input rows within a few group-times of each other and the seven
  // later ones find them in the XCD's L2 (4 MB for four clusters) instead of fetching them again from memory
  constexpr int XD = NW_ > 12 ? 4 : 8;   // input ring: block v is requested XD - 1 quads (~3.6 k cycles) before its use — the other members have pushed it out of L2
  // STRH (H = 256, a wave owns its groups): h_{t-1} is STREAMED through the operand ring behind the input blocks — [x_t | h_{t-1}]
  // is one stream of NV0 + NS sixteen-channel blocks, as in lstm_static3_kernel — instead of held in 64 registers for the
  // recurrent part.  That is what lets the H = 256 members run 16 waves (four per SIMD, 128 registers) like the H = 128 ones
  // instead of 8 (two per SIMD, 237 registers): profiles/r05/n_*.
  constexpr bool STRH = H == 256 && GS == 1 && NW_ > 8;
  static_assert((CAT ? NS0 == 1 : !(NV0 && NS0)) && (NV0 == 0 || NV0 % XD == 0) && (H == 128 || H == 256),
                "blocks of src0, or its remainder alone, or blocks of src0 + the remainder quad of src2");
  static_assert(H == 128 || NW_ <= 8 || STRH, "H = 256 with the row of h_{t-1} held: at most two waves per SIMD (register budget)");
  static_assert(!STRH || (NV0 >= XD && (NV0 + NS) % XD == 0), "streamed row: the ring depth divides the block count");
  static_assert(GS == 1 || (!SAVE && !ABLRT), "gate split: inference kernels only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ABL = ABLRT ? p.ablate : 0;

  const int b = blockIdx.x;
  const int m = (b >> 3) % NS;                                // member = hidden slice
  const int cl = ((b >> 3) / NS) * 8 + (b & 7);               // cluster (members share blockIdx & 7: one XCD, for speed only)
  if (cl >= cp.clusters_per_dir * p.ndir) return;
  const int dir = cl / cp.clusters_per_dir;
  const int ck = cl - dir * cp.clusters_per_dir;
  const int g0 = ck * cp.groups_per_cluster;
  const int g1 = g0 + cp.groups_per_cluster < p.ntasks ? g0 + cp.groups_per_cluster : p.ntasks;
  // Cooperative abort instead of a trap (include/fnssl.h, fnssl_lstm_forward): a wave whose wait runs out records a code
  // in the status word and leaves; every other wave polls that word while it waits and leaves as well; a workgroup that
  // starts late (it was not resident) sees the word at once.  What the launch has written by then is garbage that the
  // guarded fallback launch of the same call overwrites.
  if (__hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  if (cl == 0 && m == cp.stall_member) return;                // test knob: a member that never shows up

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ws = GS == 1 ? w : (w >> 2);                      // schedule lane: the wave, or its slot
  const int gate = GS == 1 ? 0 : (w & 3);                     // gate split: wave w of a slot sits on SIMD w & 3 and owns that gate
  const bool rev = dir == 1;
  const unsigned vlane = lane * 16;

  {   // my slice of the weight stream -> LDS, once
    const v4f* src = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(p.wpack[dir]) + (size_t)(m * QPS) * 4096);
    v4f* dst = reinterpret_cast<v4f*>(smem);
    for (int i = threadIdx.x; i < QPS * 4 * 64; i += NW * 64) dst[i] = src[i];
  }
  // gate split: behind the weight slice, per slot two parities of four 1-KiB gate records, then one counter per slot
  char* const xch = smem + QPS * 4096;
  unsigned* const xcnt = reinterpret_cast<unsigned*>(xch + NWS * 2 * 4 * 1024);
  if (GS > 1 && threadIdx.x < NWS) xcnt[threadIdx.x] = 0;
  __syncthreads();
  const char* const lds_rd = smem + lane * 16;
  auto rec = [&](int q, int j) { return *reinterpret_cast<const v4f*>(lds_rd + (q * 4 + j) * 1024); };
  auto rec1 = [&](int q, int j) { return *reinterpret_cast<const float*>(lds_rd + (q * 4 + j) * 1024 + 4 * gate); };   // my gate's A operand

  const unsigned st0 = (unsigned)(p.src0.st * 4), sto = (unsigned)(p.out_st * 4), stk = SUM ? (unsigned)(p.skip.st * 4) : 0u;
  const unsigned st2 = CAT ? (unsigned)(p.src2.st * 4) : 0u;
  // streaming (uni-directional layers): the host passes out - out_st, h_{-1} is the row before the chunk's first output row
  // and c_{-1} sits in the cell-state scratch, where every kernel family leaves it in the same [group][slice][lane] layout
  const unsigned cy = (unsigned)p.carry;
  unsigned* const tag_cl = cp.tags + (size_t)cl * cp.groups_per_cluster * NS;
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};

  // Deferred tag store of the previous group-step: sc1 payload stores -> s_waitcnt vmcnt(0) (every earlier vector memory
  // operation of this wave has completed) -> sc1 tag store, the "drained flag" form of MI355X_MICROARCH.md.  Deferring it to
  // the middle of the next group-step makes the drain short; measured cost against round 3's dependent-load ordering: none
  // (profiles/r04/e_*).
  unsigned pub_val = 0;
  unsigned* pub_tag = nullptr;
  auto pub_flush = [&]() {
    if (pub_tag) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(pub_tag, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pub_tag = nullptr;
    }
  };

  // ---- a group-step's addressing and its early requests (first input blocks, cell state, residual operand, tags) are
  // set up ONE group-step ahead, under the previous group-step's recurrent part: nothing of it is waited for at the top
  struct Grp {
    unsigned vo0, voo, vok, voo2, vo2 = 0;
    bool valid;
    int task, step;
  };
  // Addressing without the per-group 64-bit minimum search and division of split_addr (they cost 9 of 97 ms here, one
  // group-step every ~6 us per SIMD): the launcher has checked that offsets grow with the sequence index (so >= (q_inner - 1)
  // si, q_inner >= 16), so the wave's minimum is its first sequence, whose (qo, qi) the wave carries along incrementally;
  // a lane's distance from it is d si, plus (so - q_inner si) if the group crosses into the next outer index.
  auto locate = [&](int task, int qo0, int qi0, Grp& gr, rsrc_t& rx0, rsrc_t& ro, rsrc_t& rsk, rsrc_t& ro2, rsrc_t& rx2) {
    const int q0 = task * 16;
    gr.valid = q0 + n < p.nseq;
    const int d = q0 + n < p.nseq ? n : p.nseq - 1 - q0;
    const bool crossed = qi0 + d >= p.q_inner;
    auto one = [&](const float* base, long long so, long long si, int extra, unsigned& voff) {
      const long long delta = (long long)d * si + (crossed ? so - (long long)p.q_inner * si : 0ll);
      voff = (unsigned)(delta * 4) + (unsigned)(extra * 4);
      return make_rsrc(base + ((long long)qo0 * so + (long long)qi0 * si));
    };
    rx0 = one(p.src0.p, p.src0.so, p.src0.si, (NS0 && !CAT) ? g : 4 * g, gr.vo0);
    if constexpr (CAT) rx2 = one(p.src2.p, p.src2.so, p.src2.si, g, gr.vo2);
    ro = one(p.out, p.out_so, p.out_si, dir * H + 4 * g, gr.voo);
    gr.vok = gr.voo2 = 0;
    rsk = SUM ? one(p.skip.p, p.skip.so, p.skip.si, dir * H + 4 * g, gr.vok) : rx0;
    ro2 = SUM ? one(p.out_sum, p.out_so, p.out_si, dir * H + 4 * g, gr.voo2) : ro;
  };
  auto tt_of = [&](int step) { return (unsigned)(rev ? p.nsteps - 1 - step : step); };
  auto rc_of = [&](int task) {
    return make_rsrc(reinterpret_cast<const char*>(p.cscratch) + ((size_t)dir * (p.ntasks + 16) + task) * (NS * 1024));
  };

  if (g0 + ws >= g1) return;                                  // (a wave / slot without a group: nobody waits for it)
  unsigned long long clk0 = 0, rt0 = 0;                       // ablation twin, bit 512: the shader clock this kernel really runs at
  if (ABLRT && (ABL & 512)) {
    clk0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  bool dead = false;                                          // wave-uniform: this wave has given up (or seen that another has)
  // bounded wait for `ready()`; refreshes through `reload()`; false = gave up
  auto bounded_wait = [&](auto ready, auto reload, int sleep, unsigned code) {
    for (unsigned spins = 0; !ready(); ++spins) {
      // (gate split = the one-group-per-cluster regime: every step waits for its hand-off, so poll closely)
      if (GS > 1) __builtin_amdgcn_s_sleep(1); else if (sleep == 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(16);
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
  // Schedule of a wave within a step: its regular groups g0 + w + i NW (i < FULL / NW), then — rotate — at most one LEFTOVER
  // group g0 + FULL + j, j = (w - step) mod NW < LEFT: the groups beyond a multiple of the wave count change hands every
  // step, so every wave (and SIMD) carries the same load over the launch — config 2: 225 groups = 14 per wave + 1, the
  // fullest SIMD 56.25 group-steps per step instead of 57 (1.3 %).  A group's state lives in memory (h in the output tensor,
  // c in the scratch area, both read past the L1), so which wave of the member takes it is free.
  const int NGR = g1 - g0;
  const bool rot = cp.rotate && NGR >= 2 * NWS;
  const int FULL = rot ? (NGR / NWS) * NWS : NGR, LEFT = NGR - FULL;
  const bool single = GS == 1 && g0 + ws + NWS >= g1;                      // one group per wave: its c_t is not written yet when the next
                                                              // group-step's requests go out, so c is requested at the top
  Grp cur, nxt;
  rsrc_t rx0, ro, rsk, ro2, rx2, nrx0, nro, nrsk, nro2, nrx2;
  v4f xr[XD];
  float xs0 = 0.f;
  v4f cprev = zero4, skipv = zero4;
  unsigned tagv = 0, tagd = 0;
  auto request = [&](const Grp& gr, rsrc_t qx0, rsrc_t qsk, bool with_c, bool with_x = true) {   // early requests of group-step gr
    const unsigned tt = tt_of(gr.step);
    if (!(ABL & 8)) {
      if constexpr (NV0 > 0) if (with_x) static_for<XD>([&](auto v) { xr[v.value] = bld4(qx0, gr.vo0, tt * st0 + 64 * v.value); });
      if constexpr (NS0 > 0 && !CAT) xs0 = bld1(qx0, gr.vo0, tt * st0);
    }
    if (SUM && !(ABL & 64)) skipv = bld4(qsk, gr.vok, tt * stk + 64 * m);
    if (gr.step > 0 || cy) {
      if (gr.step > 0 && !(ABL & 256))
        tagv = __hip_atomic_load(tag_cl + (size_t)(gr.task - g0) * NS + (lane & (NS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (GS == 1 && with_c && !(ABL & 64)) cprev = bld4_l2(rc_of(gr.task), vlane, m * 1024);
    }
    if (DRIFT > 0 && !(ABL & 256) && gr.task - DRIFT * NWS >= g0)   // the members stay within DRIFT group-steps of each other (see fetch_h)
      tagd = __hip_atomic_load(tag_cl + (size_t)(gr.task - DRIFT * NWS - g0) * NS + (lane & (NS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  cur.task = g0 + ws;
  cur.step = 0;
  unsigned iter = 0;                                          // gate split: group-steps this slot has finished
  const int qo_first = (cur.task * 16) / p.q_inner, qi_first = cur.task * 16 - qo_first * p.q_inner;
  int qo0 = qo_first, qi0 = qi_first;                         // (qo, qi) of the first sequence of the NEXT group to locate
  locate(cur.task, qo0, qi0, cur, rx0, ro, rsk, ro2, rx2);
  request(cur, rx0, rsk, true);
  if constexpr (CAT) xs0 = bld1(rx2, cur.vo2, tt_of(0) * st2);

  for (;;) {
    const int step = cur.step, task = cur.task;
    const unsigned tt = tt_of(step);
    const unsigned o0 = tt * st0, oo = (tt + cy) * sto;
    const unsigned op = (rev ? tt + 1 : tt - 1 + cy) * sto;    // row of h_{step - 1}
    const rsrc_t rc = rc_of(task);
    unsigned* const tag_g = tag_cl + (size_t)(task - g0) * NS;
    if (single && (step > 0 || cy) && !(ABL & 64)) cprev = bld4_l2(rc, vlane, m * 1024);
    // (a leftover group's c_{t-1} was written by ANOTHER wave of this member, which may be behind: it is loaded in fetch_h,
    //  once the tags show that every member — this one included — has finished the group's previous step)
    // (gate split: c_{t-1} was written by the slot's gate-0 wave — same rule for every group)
    const bool late_c = GS > 1 || task >= g0 + FULL;
    v4f cprev_cur = (step > 0 || cy) ? cprev : zero4;
    const v4f skip_cur = skipv;

    // h_{step - 1} of the whole row: every member's slice, once all NS tags show it
    v4f hold[STRH ? 1 : NS];
    auto fetch_h = [&]() {
      if (DRIFT > 0 && !(ABL & 4) && !dead && task - DRIFT * NWS >= g0) {
        unsigned* const tag_d = tag_cl + (size_t)(task - DRIFT * NWS - g0) * NS;
        dead = !bounded_wait([&]() { return __builtin_amdgcn_ballot_w64(tagd < (unsigned)step + 1) == 0; },
                             [&]() { tagd = __hip_atomic_load(tag_d + (lane & (NS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, 8, 0x40000u);
      }
      if ((step > 0 || cy) && !(ABL & 16)) {
        if (step > 0 && !(ABL & 4) && !dead)
          dead = !bounded_wait([&]() { return __builtin_amdgcn_ballot_w64(tagv < (unsigned)step) == 0; },
                               [&]() { tagv = __hip_atomic_load(tag_g + (lane & (NS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, 16, 0x30000u);
        if (late_c && !(ABL & 64)) cprev_cur = bld4_l2(rc, vlane, m * 1024);
        if constexpr (!STRH) {
#pragma unroll
          for (int s = 0; s < NS; ++s) hold[s] = bld4_l2(ro, cur.voo, op + 64 * s);
        }
      } else if constexpr (!STRH) {
#pragma unroll
        for (int s = 0; s < NS; ++s) hold[s] = (ABL & 16) ? cprev : zero4;   // (ablation: any live register value)
      }
    };
    const bool hrow = step > 0 || cy;                           // (STRH) the row exists: else h_{-1} = 0
    auto hblock = [&](int j) { return hrow ? bld4_l2(ro, cur.voo, op + 64 * j) : zero4; };

    // ---- matrix phase: record j of quad Q + 1 is read from LDS right after the MFMAs that used record j of quad Q
    if (cp.prio_mode == 2) __builtin_amdgcn_s_setprio(2);
    v4f acc[4];
    v4f ra[4];
    v4f acc1 = zero4;                                          // gate split: my gate's accumulator
    float rb[4] = {0.f, 0.f, 0.f, 0.f};                        // ... and its A operands of the next quad
    if constexpr (GS == 1) {
      acc[0] = rec(0, 0);
      acc[1] = rec(0, 1);
      acc[2] = rec(0, 2);
      acc[3] = rec(0, 3);
      static_for<4>([&](auto j) { ra[j.value] = rec(1, j.value); });
    } else {
      acc1 = rec(0, gate);
      static_for<4>([&](auto j) { rb[j.value] = rec1(1, j.value); });
    }
    auto quad = [&](auto qc, float b0, float b1, float b2, float b3) {
      constexpr int Q = decltype(qc)::value;
      if constexpr (GS > 1) {   // one gate: the chain MFMA4 runs for component `gate`, record by record
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[0], b0, acc1, 0, 0, 0);
        if constexpr (Q + 1 < QPS) rb[0] = rec1(Q + 1, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[1], b1, acc1, 0, 0, 0);
        if constexpr (Q + 1 < QPS) rb[1] = rec1(Q + 1, 1);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[2], b2, acc1, 0, 0, 0);
        if constexpr (Q + 1 < QPS) rb[2] = rec1(Q + 1, 2);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[3], b3, acc1, 0, 0, 0);
        if constexpr (Q + 1 < QPS) rb[3] = rec1(Q + 1, 3);
        return;
      }
      // (the fences pin each LDS read right behind the MFMAs that used its register — 12 MFMAs before its own use; left
      //  alone the scheduler sinks the reads next to their uses and every second MFMA group waits out an LDS round trip)
      MFMA4(acc, ra[0], b0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Q + 1 < QPS) if (!(ABL & 128)) ra[0] = rec(Q + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(acc, ra[1], b1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Q + 1 < QPS) if (!(ABL & 128)) ra[1] = rec(Q + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(acc, ra[2], b2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Q + 1 < QPS) if (!(ABL & 128)) ra[2] = rec(Q + 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      MFMA4(acc, ra[3], b3);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Q + 1 < QPS) if (!(ABL & 128)) ra[3] = rec(Q + 1, 3);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto quad1 = [&](auto qc, float b0) {   // remainder quad: one record
      constexpr int Q = decltype(qc)::value;
      if constexpr (GS > 1) {
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[0], b0, acc1, 0, 0, 0);
        if constexpr (Q + 1 < QPS) static_for<4>([&](auto j) { rb[j.value] = rec1(Q + 1, j.value); });
        return;
      }
      MFMA4(acc, ra[0], b0);
      if constexpr (Q + 1 < QPS) static_for<4>([&](auto j) { ra[j.value] = rec(Q + 1, j.value); });
    };
    if constexpr (NV0 > 0) {
      static_for<NV0>([&](auto v) {
        constexpr int V = decltype(v)::value;
        if constexpr (STRH) {
          if constexpr (V == NV0 / 2) pub_flush();
          if constexpr (V == NV0 - XD) fetch_h();   // the waits: the first block of the row is requested behind this quad
        } else if constexpr (V == NV0 / 2) {   // half the input part is left to cover the recurrent operands' round trip
          pub_flush();
          fetch_h();
        }
        const v4f xb = xr[V % XD];
        quad(ic<1 + V>{}, xb.x, xb.y, xb.z, xb.w);
        if constexpr (V + XD < NV0) {
          if (!(ABL & 8)) xr[V % XD] = bld4(rx0, cur.vo0, o0 + 64 * (V + XD));
        } else if constexpr (STRH) {
          xr[V % XD] = hblock(V + XD - NV0);
        }
      });
      if constexpr (NS0 > 0) quad1(ic<1 + NV0>{}, xs0);          // the 4 concatenated data channels (src2) behind the blocks
    } else {
      pub_flush();
      fetch_h();
      quad1(ic<1>{}, xs0);
    }
    // ---- the next group-step: where it is, and its early requests (the input ring is free now)
    bool seq_next = false;                                      // the next group is this one + NW (incremental addressing)
    if (task < g0 + FULL && task + NWS < g0 + FULL) {
      nxt.task = task + NWS;
      nxt.step = step;
      seq_next = true;
    } else {
      const int j = (ws - step) & (NWS - 1);                    // (NWS is a power of two; LEFT = 0 unless rot)
      if (task < g0 + FULL && j < LEFT) {
        nxt.task = g0 + FULL + j;
        nxt.step = step;
      } else {
        nxt.task = g0 + ws;
        nxt.step = step + 1;
      }
    }
    const bool more = nxt.step < p.nsteps;
    if (more) {
      if (ABL & 1) {   // timing ablation (wrong results): every group-step uses the first group's addressing
        nxt.vo0 = cur.vo0; nxt.voo = cur.voo; nxt.vok = cur.vok; nxt.voo2 = cur.voo2; nxt.valid = cur.valid;
        nrx0 = rx0; nro = ro; nrsk = rsk; nro2 = ro2;
        if constexpr (CAT) {
          nxt.vo2 = cur.vo2;
          nrx2 = rx2;
        }
      } else {
        if (seq_next) {                                       // next group of this wave: 16 NW sequences further
          qi0 += 16 * NWS;
          while (qi0 >= p.q_inner) {
            qi0 -= p.q_inner;
            ++qo0;
          }
        } else if (nxt.step == step) {                        // a leftover group: once per step at most
          qo0 = (nxt.task * 16) / p.q_inner;
          qi0 = nxt.task * 16 - qo0 * p.q_inner;
        } else {
          qo0 = qo_first;
          qi0 = qi_first;
        }
        locate(nxt.task, qo0, qi0, nxt, nrx0, nro, nrsk, nro2, nrx2);
      }
      request(nxt, nrx0, nrsk, !single && nxt.task < g0 + FULL, !STRH);
      if constexpr (CAT) xs0 = bld1(nrx2, nxt.vo2, tt_of(nxt.step) * st2);
    }
    static_for<NS>([&](auto sp) {
      constexpr int SP = decltype(sp)::value;
      if constexpr (STRH) {
        constexpr int B = NV0 + SP, B2 = B + XD;                // block B of [x_t | h_{t-1}]; its ring slot is refilled with block B2
        const v4f hb = xr[B % XD];
        quad(ic<1 + NV0 + NS0 + SP>{}, hb.x, hb.y, hb.z, hb.w);
        if constexpr (B2 < NV0 + NS) {
          xr[B % XD] = hblock(B2 - NV0);
        } else {                                                // wrapped: the next group-step's first input blocks
          if (more) xr[B % XD] = bld4(nrx0, nxt.vo0, tt_of(nxt.step) * st0 + 64 * (B2 - NV0 - NS));
        }
      } else {
        quad(ic<1 + NV0 + NS0 + SP>{}, hold[SP].x, hold[SP].y, hold[SP].z, hold[SP].w);
      }
    });

    // ---- cell update of my 16 units, stores, publish
    if (cp.prio_mode == 2) __builtin_amdgcn_s_setprio(0);
    // A wave that gave up on a hand-off (or saw the launch draining) has multiplied garbage: it stores nothing and publishes
    // nothing — peers that are not waiting at this moment must not consume it and run on; the guarded fallback launch of the
    // same call rewrites out / out_sum / reserve (which is why no input of a call may alias them: lstm_forward_impl checks).
    if (dead) break;
    v4f cn, hn;
    if (ABL & 2) {     // timing ablation: cheap gates
      cn = acc[1] + cprev_cur + acc[0];
      hn = acc[3] + acc[2];
    } else {
      v4f ig, fg, gg, og;
      if constexpr (GS == 1) {
        ig = sigmoid4(acc[0]);
        fg = sigmoid4(acc[1]);
        gg = tanh4(acc[2]);
        og = sigmoid4(acc[3]);
      } else {
        // my gate, activated -> LDS (parity of the slot's group-step count); counter += 1; wait for all four; read the row.
        // Double buffering is enough: a wave that writes group-step i + 2's record has passed the wait of i + 1, for which
        // every sibling had written i + 1's — i.e. had finished reading i's.
        const v4f act = gate == 2 ? tanh4(acc1) : sigmoid4(acc1);
        v4f* const xb = reinterpret_cast<v4f*>(xch + (size_t)((ws * 2 + (int)(iter & 1u)) * 4) * 1024);
        xb[gate * 64 + lane] = act;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(xcnt + ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned want = 4u * (iter + 1u);
        for (unsigned spins = 0;; ++spins) {
          if (__hip_atomic_load(xcnt + ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
          // a sibling that gave up on a hand-off (or saw the launch draining) never arrives: follow it out
          if (dead || ((spins & 63u) == 63u && __hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ||
              spins > cp.spin_limit) {
            dead = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        ig = xb[lane];
        fg = xb[64 + lane];
        gg = xb[128 + lane];
        og = xb[192 + lane];
        ++iter;
      }
      cn = cell4(fg, cprev_cur, ig, gg);
      hn = mul_rn4(og, tanh4(cn));
      if constexpr (SAVE) {   // what lstm_bwd_kernel reads back (lstm_kernel.h: reserve layout), my slice of this group and step
        const rsrc_t rres = make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                                      ((size_t)dir * p.ntasks + task) * p.nsteps * (size_t)(NS * kReserveRecs * 1024));
        const unsigned rb = (tt * NS + m) * (kReserveRecs * 1024);
        bst4(ig, rres, vlane, rb);
        bst4(fg, rres, vlane, rb + 1024);
        bst4(gg, rres, vlane, rb + 2048);
        bst4(og, rres, vlane, rb + 3072);
        bst4(cn, rres, vlane, rb + 4096);
      }
    }
    asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));   // h + skip adds the ROUNDED h
    if (ABL & 32) asm volatile("" ::"v"(cn), "v"(hn));
    const bool storer = GS == 1 || gate == 0;                   // gate split: all four waves hold c_t and h_t, one stores
    if (storer && !(ABL & 32)) bst4(cn, rc, vlane, m * 1024);
    if (storer && cur.valid && !(ABL & 32)) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hn), ro, cur.voo, oo + 64 * m, 16);   // sc1: the hand-off
      if (SUM) bst4(add_rn4(hn, skip_cur), ro2, cur.voo2, oo + 64 * m);
    }
    pub_val = __builtin_amdgcn_readfirstlane((unsigned)step + 1);   // (wave-uniform: lives in an SGPR until the store)
    pub_tag = ((ABL & 256) || !storer) ? nullptr : tag_g + m;
    // gate split, or ONE group per wave: the group waits for this very hand-off at its next step — publish now, not in the
    // middle of the next input part (the deferral hides the drain behind OTHER groups' work, of which there is none here).
    // One 4-mic utterance (6 - 8 groups per cluster, every wave a single group): full-band layers 11.3 -> 10.3 ms, narrow-band
    // 16.1 -> 15.4 ms (profiles/r05/l_*).
    if (GS > 1 || single) pub_flush();
    if (!more || dead) break;
    cur = nxt;
    rx0 = nrx0;
    ro = nro;
    rsk = nrsk;
    ro2 = nro2;
    if constexpr (CAT) rx2 = nrx2;
  }
  pub_flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ABLRT && (ABL & 512) && (w == 0 || w == 15) && lane == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    printf("f32c blk %d wave %d start %llu end %llu cycles %llu MHz %.1f\n", b, w, rt0, r1, c1 - clk0,
           (double)(c1 - clk0) / (double)(r1 - rt0) * 100.0);
  }
}

template <int HH, int NV0, int NS0, int MODE, bool ABLRT = false, int DRIFT = 2, int NW_ = kF32cWaves, int GS = 1>
int launch_f32c_k(const LstmParams& p, const F32ClusterParams& cp, hipStream_t st) {
  constexpr int NS = HH / 16;
  constexpr int QPS = 1 + NV0 + NS0 + NS;
  const size_t lds = (size_t)QPS * 4096 + (GS > 1 ? (size_t)(NW_ / GS) * 8192 + 64 : 0);   // + the slots' gate records and counters
  static_assert(QPS * 4096 + (GS > 1 ? (NW_ / GS) * 8192 + 64 : 0) <= 160 * 1024, "LDS");
  auto k = lstm_f32c_kernel<HH, NV0, NS0, MODE, ABLRT, DRIFT, NW_, GS>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int ncl = cp.clusters_per_dir * p.ndir;
  const int nwg = 8 * NS * ((ncl + 7) / 8);                  // block b: XCD b & 7, member (b >> 3) % NS, cluster ((b >> 3) / NS) * 8 + (b & 7)
  // every member of every cluster must be resident at once: the grid may not exceed what the device says it can hold
  // (the waits are bounded and the call carries a guarded fallback anyway — this only avoids a launch that cannot work)
  if (!cluster_grid_fits(reinterpret_cast<const void*>(k), NW_ * 64, lds, nwg)) return kNoCluster;
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW_ * 64), lds, st, p, cp);
  FNSSL_CHECK_LAUNCH("lstm_f32c_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
