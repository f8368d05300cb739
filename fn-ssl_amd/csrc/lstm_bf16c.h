// Cluster-resident bf16 LSTM recurrence (BASELINE config 3, IPDnet's narrow-band layers): the gate rows of ONE layer are
// split over a cluster of 8 workgroups (8 CUs); every member keeps its 1/8 of the weight matrix in LDS for the whole
// launch and the cluster exchanges h_t through L2 once per step.
//
// Why (profiles/r02/f_ablate_bf16p.txt, profiles/r03/h_*): lstm_bf16p_kernel runs 64 sequences per CU against ALL gate
// rows, so every CU pulls the whole 1.06 MB matrix from L2 once per step (84 GB per launch at config 3); two launches
// on two streams take twice as long each as one alone — the L2 delivers ~11 TB/s to that pattern and that is the wall.
// Here a member owns 4 of the 32 gate-row tiles (136 records = 136 KiB of LDS, loaded once) and the cluster runs 512
// sequences against them: the per-step traffic of a CU is the B operands [x_t | h_{t-1}] of its 512 sequences
// (~0.57 MB from L2, of which x is shared by the 8 members) and no weight byte leaves the LDS again.
//
// Work split inside a member: 8 waves (two per SIMD, so that one wave's gate math runs under its partner's MFMAs
// without a hand-made software pipeline).  Wave w owns two 32-sequence tiles — tile w of each HALF of the cluster's
// batch — and alternates between them: while the h_t of one half travels to the other members, the wave works on the
// other half (a hand-off is ~2-3 us on a loaded chip; half a step is ~4 us).
//
// Hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"; nothing here depends on which XCD a member
// runs on — members of a cluster are PLACED on one XCD via blockIdx & 7 only because a same-XCD reader is faster):
//   * producer wave: its two 1-KiB operand records (blocks 2m, 2m + 1 of its sequence tile: exactly the units this
//     member computes) are stored write-through (16-byte sc1 stores); the TAG word of (half, wave, member) is stored
//     (relaxed, agent scope = sc1) only after a load issued BEHIND those stores has returned: vector memory operations
//     of a wave complete in order on gfx9, so the stores have been acknowledged by then — without draining the wave's
//     prefetch window the way s_waitcnt vmcnt(0) would;
//   * consumer wave: loads the 8 tags of its (half, wave) a few K-steps before it needs them (relaxed agent loads),
//     and only after all 8 show the step it waits for does it issue the sc1 loads of the operand records (sc1 loads
//     bypass the CU's L1, which other CUs' stores never refresh).  Tags are monotonic (step + 1) and zeroed by the
//     host before every launch; the records are double-buffered by step parity: a member can be at most one step ahead
//     of the slowest one, because step t + 1 needs every member's h_t.
//   * every wait is bounded: a tag that does not arrive within ~2 s records a code in the launch's status word and
//     traps (the launch fails loudly instead of hanging the device).  All 8 members are resident at once by
//     construction: a launch has at most one workgroup per CU (136 KiB of LDS each, <= 256 workgroups).
//
// Arithmetic: per 32-row tile the MFMA chain is the one of lstm_bf16p_kernel — ones block (bias), input blocks, recurrent
// blocks, in that order, into one fp32 accumulator — and the gate math is the same code, so the results are
// bit-identical to the pair-split kernels'.  The weight stream is fnssl_lstm_pack_bf16w's, unchanged: tile-major, so
// a member's slice is one contiguous 136 KiB chunk.
#pragma once

#include "lstm_bf16w.h"

namespace fnssl_lstm {

// (kClusterSeqs, kClusterMembers, kClusterHxBytes, kClusterTagWords: lstm_kernel.h)
constexpr unsigned kClusterSpinLimit = 1u << 20;
constexpr size_t kClusterParityBytes = kClusterHxBytes / 2;   // one cluster's records of one step parity

struct ClusterParams {
  char* hx;            // [parity 2][cluster][half 2][sequence tile 8][block 16][1 KiB]; parity 1 zeroed before the launch (h_{-1} = 0)
  unsigned parity_stride;   // bytes between the two parities = clusters of the call x kClusterParityBytes
  unsigned* tags;      // [cluster] x kClusterTagWords, zeroed before the launch
  unsigned* status;    // one word: 0 = fine
  int cl0;             // first cluster of this launch (global index over directions)
  int ncl;             // clusters in this launch
  int cl_per_dir;
  int stagger;         // experiment (FNSSL_CLUSTER_STAGGER): != 0 = every wave issues its loads at the group's first K-step
};

// ABL (make ABLATE=1 builds only; wrong results): 1 no tag waits, 2 cheap gate math, 4 no MFMAs, 8 no recurrent-operand
// loads, 16 no input loads, 32 no output stores, 64 no publish (operand stores, tag)
template <int H, int NB0, int NB2, int FLAGS, int ABL = 0>
__global__ void __launch_bounds__(512) lstm_bf16c_kernel(const LstmParams p, const ClusterParams cp) {
  constexpr int NT = H / 8, TPM = NT / kClusterMembers, NKH = H / 16, NKX = NB0 + NB2, KT = 1 + NKX + NKH;
  constexpr bool F0 = FLAGS & kW_F0, F2 = FLAGS & kW_F2, OUTF = FLAGS & kW_OUTF;
  constexpr int PUBG = (ABL & 128) ? 2 : (ABL & 256) ? 3 : 1;   // input group (K = 4 PUBG + 1) at which the previous half-step's tag is stored
  static_assert(TPM == 4 && NKH == 16 && NB0 == 16 && NB2 == 1 && !F0 && F2 && !OUTF,
                "built for IPDnet's narrow-band shape: 256 <- [256 bf16 | 16 fp32], bf16 out");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- who am I: blocks of one cluster share blockIdx & 7 (observed: one XCD)
  const int b = blockIdx.x;
  const int m = (b >> 3) & 7;
  const int cl_local = ((b >> 6) << 3) + (b & 7);
  if (cl_local >= cp.ncl) return;
  const int cg = cp.cl0 + cl_local;
  const int dir = cg / cp.cl_per_dir;
  const int cd = cg - dir * cp.cl_per_dir;

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

  // ---- per-half addressing
  unsigned vo0[2], vo2[2], voo[2];
  bool valid[2];
  long long off0[2], off2[2], offo[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    int q = cd * kClusterSeqs + hf * 256 + w * 32 + n;
    valid[hf] = q < p.nseq;
    if (q >= p.nseq) q = p.nseq - 1;
    const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;
    off0[hf] = qo * p.src0.so + qi * p.src0.si + 8 * hb;
    off2[hf] = qo * p.src2.so + qi * p.src2.si + 8 * hb;
    offo[hf] = qo * p.out_so + qi * p.out_si + dir * H + 8 * m * TPM + 16 * hb;   // (after the lane-pair swap below)
  }
  // (buffer descriptors are opaque scalars: one named variable per half, picked by the compile-time half index)
  const rsrc_t rx0_0 = split_addr_e<2>(p.src0.p, off0[0], vo0[0]), rx0_1 = split_addr_e<2>(p.src0.p, off0[1], vo0[1]);
  const rsrc_t rx2_0 = split_addr_e<4>(p.src2.p, off2[0], vo2[0]), rx2_1 = split_addr_e<4>(p.src2.p, off2[1], vo2[1]);
  const rsrc_t ro_0 = split_addr_e<2>(p.out, offo[0], voo[0]), ro_1 = split_addr_e<2>(p.out, offo[1], voo[1]);
  const unsigned st0 = (unsigned)(p.src0.st * 2), st2 = (unsigned)(p.src2.st * 4), sto = (unsigned)(p.out_st * 2);
  const rsrc_t rhx = make_rsrc(cp.hx + (size_t)cg * kClusterParityBytes);
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned vlane = lane * 16;
  unsigned* const tag_base = cp.tags + (size_t)cg * kClusterTagWords;
  // operand records of (parity, half): sequence tile w, block s
  auto hx_off = [&](int par, int hf, int s) { return (unsigned)par * cp.parity_stride + (unsigned)(((hf * 8 + w) * 16 + s) * 1024); };

  // ---- state
  v8bfw ones;
  {
    const __bf16 o1 = (__bf16)(hb == 0 ? 1.0f : 0.0f);
    ones = v8bfw{o1, o1, o1, 0, 0, 0, 0, 0};
  }
  v4f c[2][TPM];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int r = 0; r < TPM; ++r) c[hf][r] = v4f{0.f, 0.f, 0.f, 0.f};
  v8bfw win[16];                  // B operands in flight: group g (4 blocks) lives in slots 4 (g & 3) ..
  v4f sk0, sk1;                   // the fp32 input block of the coming half-step (raw)
  unsigned tagv = 0;              // tag of member (lane & 7) for the coming half-step

  const char* const lds_a = smem + lane * 16;
  auto arec = [&](int r, int k) { return __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(lds_a + (r * KT + k) * 1024)); };

  auto load_xgroup = [&](auto gc, int hf, unsigned tt) {          // input blocks 4 g .. 4 g + 3 of half hf at step tt
    constexpr int G = decltype(gc)::value;
    static_for<4>([&](auto i) {
      constexpr int B = 4 * G + decltype(i)::value;
      if constexpr (!(ABL & 16)) win[4 * (G & 3) + decltype(i)::value] = __builtin_bit_cast(v8bfw, bld4(hf ? rx0_1 : rx0_0, vo0[hf], tt * st0 + 32 * B));
    });
  };
  auto load_skip = [&](int hf, unsigned tt) {
    sk0 = bld4(hf ? rx2_1 : rx2_0, vo2[hf], tt * st2);
    sk1 = bld4(hf ? rx2_1 : rx2_0, vo2[hf], tt * st2 + 16);
  };
  auto load_hgroup = [&](auto gc, int hf, int par) {              // recurrent blocks 4 g .. 4 g + 3
    constexpr int G = decltype(gc)::value;
    static_for<4>([&](auto i) {
      constexpr int S = 4 * G + decltype(i)::value;
      if constexpr (!(ABL & 8)) win[4 * (G & 3) + decltype(i)::value] = __builtin_bit_cast(v8bfw, bld4_l2(rhx, vlane, hx_off(par, hf, S)));
    });
  };
  auto load_tags = [&](int hf) {
    tagv = __hip_atomic_load(tag_base + (hf * 8 + w) * 8 + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto tags_ready = [&](unsigned need) { return __builtin_amdgcn_ballot_w64(tagv < need) == 0; };
  auto wait_tags = [&](int hf, unsigned need) {
    if (tags_ready(need)) return;
    for (unsigned spins = 0;; ++spins) {
      __builtin_amdgcn_s_sleep(16);
      load_tags(hf);
      if (tags_ready(need)) return;
      if (spins > kClusterSpinLimit) {
        if (lane == 0) __hip_atomic_store(cp.status, 0x10000u | (unsigned)(cg & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_trap();
      }
    }
  };

  // deferred tag store of the previous half-step (see the end of half_step).  My tag words as 4-byte buffers: the store
  // is issued by all lanes at offset 4 * lane, and the descriptor's range check drops lanes 1..63 (no branch)
  const rsrc_t rtag_0 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (0 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  const rsrc_t rtag_1 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (1 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  unsigned pub_dep = 0, pub_val = 0;
  auto pub_flush = [&](int hf) {                       // hf: the half whose tag is pending
    unsigned tval = pub_val;
    asm volatile("; tag store ordered behind %1" : "+v"(tval) : "v"(pub_dep));
    __builtin_amdgcn_raw_buffer_store_b32(tval, hf ? rtag_1 : rtag_0, lane * 4, 0, 16);   // sc1: write-through
  };

  // ---- prologue: the first half-step's input groups 0..2 and its fp32 block
  const unsigned tt_first = rev ? p.nsteps - 1 : 0;
  load_xgroup(ic<0>{}, 0, tt_first);
  load_xgroup(ic<1>{}, 0, tt_first);
  load_xgroup(ic<2>{}, 0, tt_first);
  load_skip(0, tt_first);
  // drained once, with the builtin the compiler's wait-count bookkeeping sees: the step loop is then entered with nothing
  // in flight, and its header does not inherit a conservative vmcnt(0) from this path on every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt unconstrained

  // one half-step of half HF at step `step`
  auto half_step = [&](auto hfc, int step) {
    constexpr int HF = decltype(hfc)::value;
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    // the half-step after this one: the other half, same step (HF = 0) or next step (HF = 1)
    const int nstep = HF == 0 ? step : (step + 1 < p.nsteps ? step + 1 : step);
    const unsigned ttn = rev ? p.nsteps - 1 - nstep : nstep;
    // h_{step - 1} was published under parity (step - 1) & 1; step 0 reads the parity-1 records the host zeroed
    const int par = (step + 1) & 1;

    v16f acc[TPM];
    v8bfw a[TPM];
    static_for<TPM>([&](auto r) { a[decltype(r)::value] = arec(decltype(r)::value, 0); });
    static_for<KT>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      // ---- group boundaries: the group three ahead goes into the slots of the group just consumed
      if constexpr (K >= 1 && K <= NB0 && (K - 1) % 4 == 0) {                  // input groups 0..3 start at K = 1, 5, 9, 13
        constexpr int G = (K - 1) / 4;
        if constexpr (G == 0) load_xgroup(ic<3>{}, HF, tt);
        if constexpr (G == PUBG) {
          if constexpr (!(ABL & 64)) {
            if (HF == 1 || step > 0) pub_flush(HF ^ 1);
          }
        }
        if constexpr (G == 1) {
          if constexpr (!(ABL & 1)) {
            if (step > 0) wait_tags(HF, (unsigned)step);
          }
        }
        if constexpr (G >= 1) load_hgroup(ic<G - 1>{}, HF, par);              // recurrent groups 0..2
      }
      if constexpr (K >= 2 + NB0 && (K - 2 - NB0) % 4 == 0) {                  // recurrent groups 0..3 start at K = 18, 22, ..
        constexpr int G = (K - 2 - NB0) / 4;
        if constexpr (G == 0) load_hgroup(ic<3>{}, HF, par);
        if constexpr (G == 1) {
          load_xgroup(ic<0>{}, HF ^ 1, ttn);
          load_skip(HF ^ 1, ttn);                                             // (this half-step's block was used at K = 17)
        }
        if constexpr (G == 2) load_xgroup(ic<1>{}, HF ^ 1, ttn);
        if constexpr (G == 3) load_xgroup(ic<2>{}, HF ^ 1, ttn);
      }
      // ---- B operand of this K-step
      v8bfw bop;
      if constexpr (K == 0)
        bop = ones;
      else if constexpr (K <= NB0)
        bop = win[(K - 1) & 15];
      else if constexpr (K == NB0 + 1)
        bop = join8(__builtin_convertvector(sk0, v4bfw), __builtin_convertvector(sk1, v4bfw));
      else
        bop = win[(K - 2 - NB0) & 15];
      // ---- A operands one K-step ahead, 4 MFMAs.  The scheduling fences keep the four LDS reads of K-step K + 1 in
      // front of the MFMAs of K-step K: left alone the scheduler sinks each read next to its use (one MFMA ahead),
      // and every MFMA then waits out an LDS round trip (measured: 51 cycles per MFMA instead of 32)
      v8bfw an[TPM];
      if constexpr (K + 1 < KT) static_for<TPM>([&](auto r) { an[decltype(r)::value] = arec(decltype(r)::value, K + 1); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<TPM>([&](auto r) {
        constexpr int R = decltype(r)::value;
        if constexpr (ABL & 4) {
          if constexpr (K == 0) acc[R] = v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[R][K % 16] += (float)a[R][0] + (float)bop[0];
        } else if constexpr (K == 0) {
          const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[R], bop, z, 0, 0, 0);
        } else {
          acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[R], bop, acc[R], 0, 0, 0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (K + 1 < KT) static_for<TPM>([&](auto r) { a[decltype(r)::value] = an[decltype(r)::value]; });
    });

    // the tags the NEXT half-step waits for: requested now, looked at five K-steps into it
    load_tags(HF ^ 1);

    // ---- gates of my 4 tiles (units 8 (4 m + r) + 4 hb + 0..3 of sequence n), output, operand records
    const unsigned oo = tt * sto;
    v4bfw hq[TPM];
    static_for<TPM>([&](auto r) {
      constexpr int R = decltype(r)::value;
      const v16f& ac = acc[R];
      v4f cn, hn;
      if constexpr (ABL & 2) {
        cn = c[HF][R] * 0.5f + v4f{ac[0], ac[5], ac[10], ac[15]};
        hn = cn * 0.5f + v4f{ac[1], ac[6], ac[11], ac[12]};
      } else {
        const v4f ig = sigmoid4(v4f{ac[0], ac[1], ac[2], ac[3]});
        const v4f fg = sigmoid4(v4f{ac[4], ac[5], ac[6], ac[7]});
        const v4f gg = tanh4(v4f{ac[8], ac[9], ac[10], ac[11]});
        const v4f og = sigmoid4(v4f{ac[12], ac[13], ac[14], ac[15]});
        cn = cell4(fg, c[HF][R], ig, gg);
        hn = mul_rn4(og, tanh4(cn));
      }
      c[HF][R] = cn;
      hq[R] = __builtin_convertvector(hn, v4bfw);
    });
    // Output row piece of this member: units 32 m .. 32 m + 31 (64 bytes per sequence).  Lane (n, hb) holds units
    // 8 r + 4 hb + 0..3 of tile r: four 8-byte pieces 16 bytes apart.  v_permlane32_swap trades halves between lanes
    // (n, 0) and (n, 1) so that lane (n, 0) ends up with tiles 0, 1 and lane (n, 1) with tiles 2, 3, 32 contiguous bytes
    // each: two 16-byte stores instead of four 8-byte ones (the stores' acknowledgements hold back every younger load
    // of the wave — vector memory operations retire in order — and cost 2 us per step as four: profiles/r03/j_*)
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
      if constexpr (ABL & 1024) {            // timing experiment: the same bytes to a compact area (wrong results)
        __builtin_amdgcn_raw_buffer_store_b128(lo, rhx, vlane, hx_off(step & 1, HF, 2 * m), 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, rhx, vlane, hx_off(step & 1, HF, 2 * m + 1), 0);
      } else if (valid[HF]) {
        constexpr int AUX = (ABL & 512) ? 2 : (ABL & 2048) ? 16 : 0;
        __builtin_amdgcn_raw_buffer_store_b128(lo, HF ? ro_1 : ro_0, voo[HF], oo, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(hi, HF ? ro_1 : ro_0, voo[HF], oo + 16, AUX);
      }
    }
    // tiles 4 m, 4 m + 1 -> block 2 m (bytes 0-7, 8-15 of the lane's 16); tiles 4 m + 2, 4 m + 3 -> block 2 m + 1
    const int wpar = step & 1;
    if constexpr (!(ABL & 64)) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, join8(hq[0], hq[1])), rhx, vlane, hx_off(wpar, HF, 2 * m), 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, join8(hq[2], hq[3])), rhx, vlane, hx_off(wpar, HF, 2 * m + 1), 16);
    // publish, first half: a load issued behind the two stores.  Vector memory operations of a wave complete in order, so
    // once its value has arrived the stores have been acknowledged; the tag store itself follows five K-steps into the
    // next half-step (pub_flush), where waiting for that value no longer drains the wave's younger loads.
    asm volatile("" ::: "memory");
    pub_dep = __builtin_bit_cast(unsigned, bld1(rw, 0, 0));
    asm volatile("" ::: "memory");   // (keeps the load here: sunk next to its use it would make that wait a full drain)
    pub_val = (unsigned)step + 1;
    }
  };

  for (int step = 0; step < p.nsteps; ++step) {
    half_step(ic<0>{}, step);
    half_step(ic<1>{}, step);
  }
  pub_flush(1);   // (nobody waits for the last step's tags; kept so that a finished launch leaves uniform tags)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int H, int NB0, int NB2, int FLAGS, int ABL = 0>
int launch_bf16c_k(const LstmParams& p, const ClusterParams& cp, hipStream_t st) {
  constexpr int KT = 1 + NB0 + NB2 + H / 16;
  const size_t lds = (size_t)(H / 8 / kClusterMembers) * KT * 1024;
  auto k = lstm_bf16c_kernel<H, NB0, NB2, FLAGS, ABL>;
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nwg = 64 * ((cp.ncl + 7) / 8);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, st, p, cp);
  FNSSL_CHECK_LAUNCH("lstm_bf16c_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
