// Cluster-resident bf16 LSTM recurrence (BASELINE config 3: IPDnet's narrow-band and full-band layers): the gate rows of
// ONE layer are split over a cluster of 8 (H = 256) or 4 (H = 128) workgroups = CUs; every member keeps its slice of the
// weight matrix in LDS for the whole launch and the cluster exchanges h_t through L2 once per step.
//
// Why (profiles/r02/f_ablate_bf16p.txt, profiles/r03/h_*, j_*): lstm_bf16p_kernel runs 64 sequences per CU against ALL
// gate rows, so every CU pulls the whole matrix (1.06 MB at H = 256) from L2 once per step — 88 GB per launch at config 3,
// and the L2 hands that pattern ~11 TB/s.  Here a member owns 4 of the gate-row tiles (136 / 104 / 40 KiB of LDS, loaded
// once) and the cluster runs 512 / 768 sequences against them: the per-step traffic of a CU is the B operands
// [x_t | h_{t-1}] of those sequences (0.57 MB at H = 256, of which x is shared by the members: 43 GB per launch) and no
// weight byte leaves the LDS again.
//
// Work split inside a member: 8 waves, two per SIMD.  Wave w owns one 32-sequence tile — tile w — of each of the
// cluster's NP "parts" and cycles through them: while the h_t of one part travels to the other members, the wave works on
// the next part (a hand-off is ~2-3 us on a loaded chip; a part takes ~8 us).  Measured (k_pmc_sq_*): the two waves of a
// SIMD mostly take turns — MFMAs 55 % of the cycles, gate math most of the rest, both at once 9 %.
//
// Hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"; nothing here depends on which XCD a member
// runs on — members of a cluster are PLACED on one XCD via blockIdx & 7 only because a same-XCD reader is faster;
// FNSSL_CLUSTER_SPREAD=1 places them on different XCDs: same bits, 14 % slower):
//   * producer wave: its two 1-KiB operand records (blocks 2m, 2m + 1 of its sequence tile: exactly the units this
//     member computes) are stored write-through (16-byte sc1 stores); the TAG word of (part, wave, member) is stored
//     (sc1) behind an s_waitcnt vmcnt(0), five K-steps into the next part (round 3 ordered it behind a dependent load
//     instead; the explicit drain costs nothing measurable: profiles/r04/);
//   * consumer wave: loads the tags of its (part, wave) at the end of the previous part (relaxed agent loads), looks at
//     them a few K-steps later, and only after all members show the step it waits for does it issue the sc1 loads of
//     the operand records (sc1 loads bypass the CU's L1, which other CUs' stores never refresh).  Tags are monotonic
//     (step + 1) and zeroed by the host before every call; the records are double-buffered by step parity: a member can be
//     at most one step ahead of the slowest one, because step t + 1 needs every member's h_t; the parity-1 records are
//     zeroed by the host, so step 0 reads h_{-1} = 0 like any other step;
//   * every wait is bounded and cooperative: a tag that does not arrive within the spin limit (~1.5 s) records a code in
//     the call's status word (fnssl_lstm_cluster_status); the wave stops waiting, every other wave sees the word at its next
//     poll, all workgroups drain — no trap — and the pair-split kernels the same call has enqueued behind this one (guarded
//     by that word) recompute the layer.  The launcher checks co-residency of the grid with the occupancy query first
//     (kNoCluster: the caller takes the pair-split kernels unguarded).
//
// Arithmetic: per 32-row tile the MFMA chain is the one of lstm_bf16p_kernel — ones block (bias), input blocks, recurrent
// blocks, in that order, into one fp32 accumulator — and the gate math is the same code, so the results are
// bit-identical to the pair-split kernels'.  The weight stream is fnssl_lstm_pack_bf16w's, unchanged: tile-major, so
// a member's slice is one contiguous chunk.
#pragma once

#include "lstm_bf16w.h"

namespace fnssl_lstm {

// Geometry per hidden size (cluster_geom in lstm_kernel.h has the host-side copy used for the workspace):
//   H = 256 (narrow-band): 8 members x 4 tiles, 2 parts  -> 512 sequences per cluster
//   H = 128 (full-band):   4 members x 4 tiles, 3 parts  -> 768 sequences per cluster (config 3: 19200 sequences per
//                          direction = exactly 25 clusters, 50 clusters x 4 CUs in ONE launch; with 2 parts it would be
//                          76 clusters = 304 workgroups, i.e. two rounds)
// "part" = one 32-sequence tile per wave; a wave cycles through its parts, so a part's h_t has (parts - 1) part-times
// to reach the other members.
// Round 6: the tiles per cluster are a launch parameter (ClusterParams::tpc <= 8 NP).  Tile t = 8 pt + w is part pt of wave w;
// a wave runs the parts it has (at least two: with fewer than two live tiles the second is a phantom, as every tile past the
// end of the batch always was), a wave without any leaves after the weight load.  Config 3's full-band layers (1200 tiles)
// ran as 50 clusters x 24 tiles on 200 CUs = 3 + 3 parts per SIMD; the launcher now cuts them into 60 clusters x 20 tiles
// (240 CUs): waves 0-3 three parts, waves 4-7 two = 5 per SIMD (forward_bf16c picks the split by that count).

struct ClusterParams {
  char* hx;                 // [parity 2][cluster][part][sequence tile 8][block H/16][1 KiB]; parity 1 zeroed before the launch (h_{-1} = 0)
  unsigned parity_stride;   // bytes between the two parities = clusters of the call x bytes per cluster and parity
  unsigned* tags;           // [cluster] x kClusterTagWords, zeroed before the launch
  unsigned* status;         // one word: 0 = fine
  int cl0;                  // first cluster of this launch (global index over directions)
  int ncl;                  // clusters in this launch
  int cl_per_dir;
  int tpc;                  // 32-sequence tiles per cluster (<= 8 NP): cluster cd of a direction owns sequences [cd * 32 tpc, (cd + 1) * 32 tpc)
  int spread;               // test knob (FNSSL_CLUSTER_SPREAD=1): members of a cluster = CONSECUTIVE blocks, i.e. different XCDs
  unsigned spin_limit;      // spins (~1.5 us each) a wave waits for a tag before it gives up (cluster_spin_limit())
  int stall_member;         // test knob: this member of the call's first cluster exits at once (-1: none)
};

// ABL (make ABLATE=1 builds only; wrong results): 1 no tag waits, 2 cheap gate math, 8 no recurrent-operand loads,
// 16 no input loads, 32 no output stores, 64 no publish (operand stores, tag)
template <int H, int NB0, int NB2, int FLAGS, int ABL = 0, int WG_ = 0, int AD = 1>
__global__ void __launch_bounds__(512) lstm_bf16c_kernel(const LstmParams p, const ClusterParams cp) {
  constexpr int CL = cluster_members(H), NP = cluster_parts(H);
  constexpr int NT = H / 8, TPM = NT / CL, NKH = H / 16, NKX = NB0 + NB2, KT = 1 + NKX + NKH;
  constexpr bool F0 = FLAGS & kW_F0, F2 = FLAGS & kW_F2, OUTF = FLAGS & kW_OUTF;
  // B-operand window: groups of 4 blocks; GX input + GH recurrent groups per part; WG groups of registers; the group
  // WG - 1 ahead is requested when a group starts, into the slots of the group consumed before it
  constexpr int GX = NB0 / 4, GH = NKH / 4, GP = GX + GH;
  // (GX = 0 — block 1's full-band layer, whose only input is the fp32 block: the window is the part's recurrent groups,
  //  requested at the end of the PREVIOUS part's matrix phase, before its gate math)
  constexpr int WG = GX == 0 ? GH : WG_ ? WG_ : (GP % 4 == 0 ? 4 : 3), WS = 4 * WG;   // AD: K-steps the A operands are read ahead
  static_assert(TPM == 4 && NB0 % 4 == 0 && NKH % 4 == 0 && NB2 == 1 && !F0 && F2 && !OUTF,   // (block 1: the launcher passes the fp32 input as src2)
                "built for IPDnet's shapes: [16 n bf16 channels | 16 fp32 channels] in, bf16 out");
  static_assert(GP % WG == 0 && (GX == 0 || (WG - 1 <= GX && GX >= 2)), "window groups must tile a part; the look-ahead stays inside the next part's input groups");
  constexpr int GTAG = GX - (WG - 1);                   // group at whose start the first recurrent group is requested
  static_assert(GX == 0 || GTAG >= 1, "the tag store of the previous part (group 1) precedes the tag wait");
  static_assert(GX > 0 || GH >= 2, "the tag store of the previous part sits at the second recurrent group");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- who am I: blocks of one cluster share blockIdx & 7 (observed: one XCD)
  const int b = blockIdx.x;
  const int m = cp.spread ? b % CL : (b >> 3) % CL;
  const int cl_local = cp.spread ? b / CL : ((b >> 3) / CL) * 8 + (b & 7);
  if (cl_local >= cp.ncl) return;
  const int cg = cp.cl0 + cl_local;
  const int dir = cg / cp.cl_per_dir;
  const int cd = cg - dir * cp.cl_per_dir;
  // Cooperative abort instead of a trap (include/fnssl.h, fnssl_lstm_forward): a wave whose wait runs out records a code in
  // the status word and leaves at the end of its step; every other wave polls the word while it waits; a workgroup that
  // starts late sees it at once.  The guarded pair-split launch of the same call then recomputes the layer.
  if (__hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  if (cg == 0 && m == cp.stall_member) return;           // test knob: a member that never shows up

  const int lane = threadIdx.x & 63;
  const int n = lane & 31, hb = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool rev = dir == 1;

  // ---- my slice of the weight stream -> LDS, once
  {
    const v4f* src = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(p.wpack[dir]) + (size_t)(m * TPM * KT) * 1024);
    v4f* dst = reinterpret_cast<v4f*>(smem);
    for (int i = threadIdx.x; i < TPM * KT * 64; i += 512) dst[i] = src[i];
  }
  __syncthreads();

  // ---- per-part addressing (buffer descriptors are opaque scalars: named variables, picked by the part index)
  unsigned vo0[3], vo2[3], voo[3];
  bool valid[3];
  long long off0[3], off2[3], offo[3];
  int npresent = 0;               // parts of this wave that hold at least one live sequence (a prefix of 0 .. NP - 1)
#pragma unroll
  for (int pt = 0; pt < 3; ++pt) {
    const int tile = (pt < NP ? pt : 0) * 8 + w;
    int q = cd * (cp.tpc * 32) + tile * 32 + n;
    const bool present = pt < NP && tile < cp.tpc && cd * (cp.tpc * 32) + tile * 32 < p.nseq;   // wave-uniform
    npresent += present ? 1 : 0;
    valid[pt] = present && q < p.nseq;
    if (q >= p.nseq) q = p.nseq - 1;
    const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;
    off0[pt] = qo * p.src0.so + qi * p.src0.si + 8 * hb;
    off2[pt] = qo * p.src2.so + qi * p.src2.si + 8 * hb;
    offo[pt] = qo * p.out_so + qi * p.out_si + dir * H + 8 * m * TPM + 16 * hb;   // (after the lane-pair swap below)
  }
  const rsrc_t rx0_0 = split_addr_e<2>(p.src0.p, off0[0], vo0[0]), rx0_1 = split_addr_e<2>(p.src0.p, off0[1], vo0[1]),
               rx0_2 = split_addr_e<2>(p.src0.p, off0[2], vo0[2]);
  const rsrc_t rx2_0 = split_addr_e<4>(p.src2.p, off2[0], vo2[0]), rx2_1 = split_addr_e<4>(p.src2.p, off2[1], vo2[1]),
               rx2_2 = split_addr_e<4>(p.src2.p, off2[2], vo2[2]);
  const rsrc_t ro_0 = split_addr_e<2>(p.out, offo[0], voo[0]), ro_1 = split_addr_e<2>(p.out, offo[1], voo[1]),
               ro_2 = split_addr_e<2>(p.out, offo[2], voo[2]);
  auto RX0 = [&](int pt) { return pt == 0 ? rx0_0 : pt == 1 ? rx0_1 : rx0_2; };
  auto RX2 = [&](int pt) { return pt == 0 ? rx2_0 : pt == 1 ? rx2_1 : rx2_2; };
  auto RO = [&](int pt) { return pt == 0 ? ro_0 : pt == 1 ? ro_1 : ro_2; };
  const unsigned st0 = (unsigned)(p.src0.st * 2), st2 = (unsigned)(p.src2.st * 4), sto = (unsigned)(p.out_st * 2);
  // a wave without a live tile has nothing to publish and nobody waits for it (presence depends on (cluster, part, wave) only,
  // the same in every member); no workgroup barrier follows the weight load
  if (__builtin_amdgcn_readfirstlane(npresent) == 0) return;
  const rsrc_t rhx = make_rsrc(cp.hx + (size_t)cg * cluster_parity_bytes(H));
  const unsigned vlane = lane * 16;
  unsigned* const tag_base = cp.tags + (size_t)cg * kClusterTagWords;
  // operand records of (parity, part): sequence tile w, block s
  auto hx_off = [&](int par, int pt, int s) { return (unsigned)par * cp.parity_stride + (unsigned)(((pt * 8 + w) * NKH + s) * 1024); };

  // ---- state
  v8bfw ones;
  {
    const __bf16 o1 = (__bf16)(hb == 0 ? 1.0f : 0.0f);
    ones = v8bfw{o1, o1, o1, 0, 0, 0, 0, 0};
  }
  v4f c[NP][TPM];
#pragma unroll
  for (int pt = 0; pt < NP; ++pt)
#pragma unroll
    for (int r = 0; r < TPM; ++r) c[pt][r] = v4f{0.f, 0.f, 0.f, 0.f};
  v8bfw win[WS];                  // B operands in flight: window group g (4 blocks) lives in slots 4 (g % WG) ..
  v4f sk0, sk1;                   // the fp32 input block of the coming part (raw)
  unsigned tagv = 0;              // tag of member (lane % CL) for the coming part

  const char* const lds_a = smem + lane * 16;
  auto arec = [&](int r, int k) { return __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(lds_a + (r * KT + k) * 1024)); };

  // window group T of part pt (T < GX: input blocks 4 T ..; else recurrent blocks 4 (T - GX) ..) into its slots
  auto load_group = [&](auto tc, int pt, unsigned tt, int par) {
    constexpr int T = decltype(tc)::value;
    static_for<4>([&](auto i) {
      constexpr int I = decltype(i)::value;
      if constexpr (T < GX) {
        if constexpr (!(ABL & 16)) win[4 * (T % WG) + I] = __builtin_bit_cast(v8bfw, bld4(RX0(pt), vo0[pt], tt * st0 + 32 * (4 * T + I)));
      } else {
        if constexpr (!(ABL & 8)) win[4 * (T % WG) + I] = __builtin_bit_cast(v8bfw, bld4_l2(rhx, vlane, hx_off(par, pt, 4 * (T - GX) + I)));
      }
    });
  };
  auto load_skip = [&](int pt, unsigned tt) {
    sk0 = bld4(RX2(pt), vo2[pt], tt * st2);
    sk1 = bld4(RX2(pt), vo2[pt], tt * st2 + 16);
  };
  auto load_tags = [&](int pt) {
    tagv = __hip_atomic_load(tag_base + (pt * 8 + w) * 8 + (lane % CL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto tags_ready = [&](unsigned need) { return __builtin_amdgcn_ballot_w64(tagv < need) == 0; };
  bool dead = false;              // wave-uniform: this wave has given up (or seen that another one has)
  auto wait_tags = [&](int pt, unsigned need) {
    if (dead || tags_ready(need)) return;
    for (unsigned spins = 0;; ++spins) {
      __builtin_amdgcn_s_sleep(16);
      load_tags(pt);
      if (tags_ready(need)) return;
      if (spins > cp.spin_limit) {
        if (lane == 0) __hip_atomic_store(cp.status, 0x10000u | (unsigned)(cg & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        return;
      }
      if ((spins & 63) == 63 && __hip_atomic_load(cp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        dead = true;
        return;
      }
    }
  };

  // deferred tag store of the previous part (see the end of part_step).  My tag words as 4-byte buffers: the store is
  // issued by all lanes at offset 4 * lane, and the descriptor's range check drops lanes 1..63 (no branch)
  const rsrc_t rtag_0 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (0 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  const rsrc_t rtag_1 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (1 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  const rsrc_t rtag_2 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (2 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  unsigned pub_val = 0;
  auto pub_flush = [&](int pt) {                       // pt: the part whose tag is pending
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // release: every earlier vector memory operation of this wave has completed
    const unsigned tval = pub_val;
    __builtin_amdgcn_raw_buffer_store_b32(tval, pt == 0 ? rtag_0 : pt == 1 ? rtag_1 : rtag_2, lane * 4, 0, 16);   // sc1: write-through
  };

  // ---- prologue: the first part's input groups 0 .. WG - 2 (GX = 0: its recurrent groups = the zeroed h_{-1}) and its fp32 block
  const unsigned tt_first = rev ? p.nsteps - 1 : 0;
  if constexpr (GX > 0)
    static_for<WG - 1>([&](auto g) { load_group(g, 0, tt_first, 0); });
  else
    static_for<GH>([&](auto g) { load_group(g, 0, tt_first, 1); });
  load_skip(0, tt_first);
  // drained once, with the builtin the compiler's wait-count bookkeeping sees: the step loop is then entered with nothing
  // in flight, and its header does not inherit a conservative vmcnt(0) from this path on every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt unconstrained

  // one part (32 sequences per wave) at step `step`
  auto part_step = [&](auto ptc, auto npwc, int step) {
    constexpr int PT = decltype(ptc)::value, NPW = decltype(npwc)::value;   // NPW: parts this wave cycles through
    constexpr int PN = (PT + 1) % NPW, PP = (PT + NPW - 1) % NPW;     // the part after / before this one
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    // the part after this one: same step, or the next step's part 0
    const int nstep = PT + 1 < NPW ? step : (step + 1 < p.nsteps ? step + 1 : step);
    const unsigned ttn = rev ? p.nsteps - 1 - nstep : nstep;
    // h_{step - 1} was published under parity (step - 1) & 1; step 0 reads the parity-1 records the host zeroed
    const int par = (step + 1) & 1;

    v16f acc[TPM];
    v8bfw ar[AD + 1][TPM];                                            // A operands of K-steps K .. K + AD (rotating)
    __builtin_amdgcn_s_setprio(2);   // the matrix phase outranks the SIMD partner's gate phase at issue (measured: 1-2 %)
    static_for<AD>([&](auto d) { static_for<TPM>([&](auto r) { ar[decltype(d)::value][decltype(r)::value] = arec(decltype(r)::value, decltype(d)::value); }); });
    static_for<KT>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      // window position of this K-step's operand: input block K - 1, or NB0 + recurrent block; -1: ones / fp32 block
      constexpr int POS = (K >= 1 && K <= NB0) ? K - 1 : (K >= NB0 + 2 ? K - 2 : -1);
      if constexpr (GX == 0) {
        if constexpr (K == 0) load_tags(PN);             // looked at after this part's matrix phase
        if constexpr (POS == 4) {
          if constexpr (!(ABL & 64)) {
            if (PT > 0 || step > 0) pub_flush(PP);
          }
        }
      } else if constexpr (POS >= 0 && POS % 4 == 0) {   // a window group starts: request the group WG - 1 ahead
        constexpr int G = POS / 4, T = G + WG - 1;
        if constexpr (G == 1) {
          if constexpr (!(ABL & 64)) {
            if (PT > 0 || step > 0) pub_flush(PP);
          }
        }
        if constexpr (G == GTAG) {
          if constexpr (!(ABL & 1)) {
            if (step > 0) wait_tags(PT, (unsigned)step);
          }
        }
        if constexpr (T < GP)
          load_group(ic<T>{}, PT, tt, par);
        else
          load_group(ic<T - GP>{}, PN, ttn, 0);           // (an input group of the next part)
        if constexpr (G == GX) load_skip(PN, ttn);        // (this part's fp32 block was used at K = NB0 + 1)
      }
      // ---- B operand of this K-step
      v8bfw bop;
      if constexpr (K == 0)
        bop = ones;
      else if constexpr (K == NB0 + 1)
        bop = join8(__builtin_convertvector(sk0, v4bfw), __builtin_convertvector(sk1, v4bfw));
      else
        bop = win[POS % WS];
      // ---- A operands AD K-steps ahead, 4 MFMAs.  The scheduling fences keep the four LDS reads of K-step K + AD in
      // front of the MFMAs of K-step K (left alone the scheduler sinks each read next to its use, one MFMA ahead)
      if constexpr (K + AD < KT) static_for<TPM>([&](auto r) { ar[(K + AD) % (AD + 1)][decltype(r)::value] = arec(decltype(r)::value, K + AD); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<TPM>([&](auto r) {
        constexpr int R = decltype(r)::value;
        if constexpr (K == 0) {
          const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[K % (AD + 1)][R], bop, z, 0, 0, 0);
        } else {
          acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[K % (AD + 1)][R], bop, acc[R], 0, 0, 0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });

    __builtin_amdgcn_s_setprio(0);
    if constexpr (GX > 0) {
      // the tags the NEXT part waits for: requested now, looked at a few K-steps into it
      load_tags(PN);
    } else {
      // the NEXT part's recurrent operands and fp32 block: requested now, they land under this part's gate math
      if (PT + 1 < NPW || step + 1 < p.nsteps) {
        if constexpr (!(ABL & 1)) {
          if (nstep > 0) wait_tags(PN, (unsigned)nstep);
        }
        static_for<GH>([&](auto g) { load_group(g, PN, ttn, (nstep + 1) & 1); });
      }
      load_skip(PN, ttn);
    }

    // ---- gates of my 4 tiles (units 8 (4 m + r) + 4 hb + 0..3 of sequence n), output, operand records
    const unsigned oo = tt * sto;
    v4bfw hq[TPM];
    static_for<TPM>([&](auto r) {
      constexpr int R = decltype(r)::value;
      const v16f& ac = acc[R];
      v4f cn, hn;
      if constexpr (ABL & 2) {
        cn = c[PT][R] * 0.5f + v4f{ac[0], ac[5], ac[10], ac[15]};
        hn = cn * 0.5f + v4f{ac[1], ac[6], ac[11], ac[12]};
      } else {
        const v4f ig = sigmoid4(v4f{ac[0], ac[1], ac[2], ac[3]});
        const v4f fg = sigmoid4(v4f{ac[4], ac[5], ac[6], ac[7]});
        const v4f gg = tanh4(v4f{ac[8], ac[9], ac[10], ac[11]});
        const v4f og = sigmoid4(v4f{ac[12], ac[13], ac[14], ac[15]});
        cn = cell4(fg, c[PT][R], ig, gg);
        hn = mul_rn4(og, tanh4(cn));
      }
      c[PT][R] = cn;
      hq[R] = __builtin_convertvector(hn, v4bfw);
    });
    // Output row piece of this member: units 32 m .. 32 m + 31 (64 bytes per sequence).  Lane (n, hb) holds units
    // 8 r + 4 hb + 0..3 of tile r: four 8-byte pieces 16 bytes apart.  v_permlane32_swap trades halves between lanes
    // (n, 0) and (n, 1) so that lane (n, 0) ends up with tiles 0, 1 and lane (n, 1) with tiles 2, 3, 32 contiguous bytes
    // each: two 16-byte stores instead of four 8-byte ones (the stores' acknowledgements hold back every younger load
    // of the wave — vector memory operations retire in order: profiles/r03/j_*)
    if constexpr (!(ABL & 32)) {
      const v2u d0 = __builtin_bit_cast(v2u, hq[0]), d1 = __builtin_bit_cast(v2u, hq[1]);
      const v2u d2 = __builtin_bit_cast(v2u, hq[2]), d3 = __builtin_bit_cast(v2u, hq[3]);
      // swap(a, b): first result = {lanes 0-31: a's, lanes 32-63: b's lower half}; second = {a's upper half, b's}
      const auto s00 = __builtin_amdgcn_permlane32_swap(d0[0], d2[0], false, false);
      const auto s01 = __builtin_amdgcn_permlane32_swap(d0[1], d2[1], false, false);
      const auto s10 = __builtin_amdgcn_permlane32_swap(d1[0], d3[0], false, false);
      const auto s11 = __builtin_amdgcn_permlane32_swap(d1[1], d3[1], false, false);
      const v4u lo = {s00[0], s01[0], s00[1], s01[1]};   // hb 0: tile 0 units 0-3, 4-7; hb 1: tile 2
      const v4u hi = {s10[0], s11[0], s10[1], s11[1]};   // hb 0: tile 1;                hb 1: tile 3
      // (a wave that gave up on a hand-off, or saw the launch draining, has multiplied operands it did not wait for: it stores
      //  and publishes nothing more — the guarded fallback launch of the same call rewrites out / out_sum)
      if (valid[PT] && !dead) {
        __builtin_amdgcn_raw_buffer_store_b128(lo, RO(PT), voo[PT], oo, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, RO(PT), voo[PT], oo + 16, 0);
      }
    }
    // tiles 4 m, 4 m + 1 -> block 2 m (bytes 0-7, 8-15 of the lane's 16); tiles 4 m + 2, 4 m + 3 -> block 2 m + 1
    const int wpar = step & 1;
    if constexpr (!(ABL & 64)) if (!dead) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, join8(hq[0], hq[1])), rhx, vlane, hx_off(wpar, PT, 2 * m), 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, join8(hq[2], hq[3])), rhx, vlane, hx_off(wpar, PT, 2 * m + 1), 16);
      // publish: the tag store follows five K-steps into the next part (pub_flush), behind an s_waitcnt vmcnt(0) — by then
      // the two stores above have long been acknowledged and the drain is short
      pub_val = (unsigned)step + 1;
    }
  };

  auto run = [&](auto npwc) {
    constexpr int NPW = decltype(npwc)::value;
    for (int step = 0; step < p.nsteps && !dead; ++step) static_for<NPW>([&](auto pt) { part_step(pt, npwc, step); });
    if (!dead) pub_flush(NPW - 1);   // (nobody waits for the last step's tags; kept so that a finished launch leaves uniform tags)
  };
  if constexpr (NP == 3) {
    if (__builtin_amdgcn_readfirstlane(npresent) == 3)
      run(ic<3>{});
    else
      run(ic<2>{});   // two live tiles — or one and a phantom (stores and publishes nothing that anybody reads)
  } else {
    run(ic<NP>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int H, int NB0, int NB2, int FLAGS, int ABL = 0, int WG_ = 0, int AD = 1>
int launch_bf16c_k(const LstmParams& p, const ClusterParams& cp, hipStream_t st) {
  constexpr int KT = 1 + NB0 + NB2 + H / 16, CL = cluster_members(H);
  const size_t lds = (size_t)(H / 8 / CL) * KT * 1024;
  auto k = lstm_bf16c_kernel<H, NB0, NB2, FLAGS, ABL, WG_, AD>;
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nwg = 8 * CL * ((cp.ncl + 7) / 8);
  // every member of every cluster of the launch must be resident at once (see launch_f32c_k)
  if (!cluster_grid_fits(reinterpret_cast<const void*>(k), 512, lds, nwg)) return kNoCluster;
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, st, p, cp);
  FNSSL_CHECK_LAUNCH("lstm_bf16c_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
