// Cluster-resident bf16 LSTM recurrence, 64 sequences per wave (second formulation of lstm_bf16c.h; same cluster
// geometry, same hand-off protocol and workspace layout, same bits).
//
// What lstm_bf16c.h left on the table (profiles/r03/j_*): its waves own 32 sequences, so every 1-KiB A record is read
// from LDS once per MFMA, and two waves share a SIMD so that one's gate math hides under the other's MFMAs — the matrix
// pipes end up 41 % busy.  Here a member runs FOUR waves, one per SIMD, 512 registers each:
//   * a wave owns TWO 32-sequence tiles of every part: one A record feeds two MFMAs (half the LDS reads per flop), a
//     K-step is 8 MFMAs = 256 cycles of matrix pipe behind 4 LDS reads;
//   * two accumulator sets (2 x 128 registers): while part p accumulates into one, the gate math of part p - 1 reads the
//     other — cut into 24 chunks of ~40 vector instructions that are issued BETWEEN the MFMAs of the first 24 K-steps
//     (an MFMA occupies the matrix pipe for 32 cycles and the wave's issue slot for 4);
//   * the previous part's outputs and operand records are stored at K-steps 12 and 24, the tag follows at K-step 30,
//     so a hand-off has more than a part-time left before the consumer looks at the tag (K-step 13 of the part after).
// Everything else — member / tile / record layout, tags, parity, bounded waits — is lstm_bf16c.h's.
#pragma once

#include "lstm_bf16c.h"

namespace fnssl_lstm {

template <int H, int NB0, int NB2, int FLAGS, int ABL = 0>
__global__ void __launch_bounds__(256) lstm_bf16d_kernel(const LstmParams p, const ClusterParams cp) {
  constexpr int CL = cluster_members(H), NP = cluster_parts(H), NJ = 2;
  constexpr int NT = H / 8, TPM = NT / CL, NKH = H / 16, NKX = NB0 + NB2, KT = 1 + NKX + NKH;
  constexpr bool F0 = FLAGS & kW_F0, F2 = FLAGS & kW_F2, OUTF = FLAGS & kW_OUTF;
  constexpr int GX = NB0 / 4, GH = NKH / 4, GP = GX + GH;
  constexpr int WG = 2, WS = 4 * WG;                    // operand window: two groups of 4 blocks (x 2 tiles)
  static_assert(NP == 2 && TPM == 4 && NB0 % 4 == 0 && NKH % 4 == 0 && NB2 == 1 && !F0 && F2 && !OUTF && GX >= 2 && GP % WG == 0,
                "built for the two-part shapes of lstm_bf16c.h");
  constexpr int GTAG = GX - 1;                          // group at whose start the first recurrent group is requested
  constexpr int KG = 3 * TPM * NJ;                      // K-steps that carry a gate chunk (3 chunks per tile)
  constexpr int KPUB = KG + 6;                          // K-step of the deferred tag store
  static_assert(KPUB < KT && KG + 2 <= KT, "the previous part's epilogue fits under this part's matrix phase");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int b = blockIdx.x;
  const int m = (b >> 3) % CL;
  const int cl_local = ((b >> 3) / CL) * 8 + (b & 7);
  if (cl_local >= cp.ncl) return;
  const int cg = cp.cl0 + cl_local;
  const int dir = cg / cp.cl_per_dir;
  const int cd = cg - dir * cp.cl_per_dir;

  const int lane = threadIdx.x & 63;
  const int n = lane & 31, hb = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool rev = dir == 1;

  {
    const v4f* src = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(p.wpack[dir]) + (size_t)(m * TPM * KT) * 1024);
    v4f* dst = reinterpret_cast<v4f*>(smem);
    for (int i = threadIdx.x; i < TPM * KT * 64; i += 256) dst[i] = src[i];
  }
  __syncthreads();

  // ---- addressing: ONE descriptor per tensor (base = the smallest offset of the wave's four tiles), a byte offset per
  // (part, tile) and lane; the launcher checks that a cluster's sequences span < 4 GB
  long long off0[NP][NJ], off2[NP][NJ], offo[NP][NJ];
  bool valid[NP][NJ];
  long long mn0 = 0x7fffffffffffffffll, mn2 = mn0, mno = mn0;
#pragma unroll
  for (int pt = 0; pt < NP; ++pt)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int q = cd * (NP * 256) + pt * 256 + (NJ * w + j) * 32 + n;
      valid[pt][j] = q < p.nseq;
      if (q >= p.nseq) q = p.nseq - 1;
      const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;
      off0[pt][j] = qo * p.src0.so + qi * p.src0.si + 8 * hb;
      off2[pt][j] = qo * p.src2.so + qi * p.src2.si + 8 * hb;
      offo[pt][j] = qo * p.out_so + qi * p.out_si + dir * H + 8 * m * TPM + 16 * hb;
      mn0 = off0[pt][j] < mn0 ? off0[pt][j] : mn0;
      mn2 = off2[pt][j] < mn2 ? off2[pt][j] : mn2;
      mno = offo[pt][j] < mno ? offo[pt][j] : mno;
    }
  auto wave_min = [&](long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const long long o = __shfl_xor(v, d, 64);
      v = o < v ? o : v;
    }
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffll));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  mn0 = wave_min(mn0);
  mn2 = wave_min(mn2);
  mno = wave_min(mno);
  const rsrc_t rx0 = make_rsrc(reinterpret_cast<const char*>(p.src0.p) + mn0 * 2);
  const rsrc_t rx2 = make_rsrc(reinterpret_cast<const char*>(p.src2.p) + mn2 * 4);
  const rsrc_t ro = make_rsrc(reinterpret_cast<const char*>(p.out) + mno * 2);
  unsigned vo0[NP][NJ], vo2[NP][NJ], voo[NP][NJ];
#pragma unroll
  for (int pt = 0; pt < NP; ++pt)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      vo0[pt][j] = (unsigned)((off0[pt][j] - mn0) * 2);
      vo2[pt][j] = (unsigned)((off2[pt][j] - mn2) * 4);
      voo[pt][j] = (unsigned)((offo[pt][j] - mno) * 2);
    }
  const unsigned st0 = (unsigned)(p.src0.st * 2), st2 = (unsigned)(p.src2.st * 4), sto = (unsigned)(p.out_st * 2);
  const rsrc_t rhx = make_rsrc(cp.hx + (size_t)cg * cluster_parity_bytes(H));
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned vlane = lane * 16;
  unsigned* const tag_base = cp.tags + (size_t)cg * kClusterTagWords;
  // operand records of (parity, part): sequence tile NJ w + j, block s
  auto hx_off = [&](int par, int pt, int j, int s) {
    return (unsigned)par * cp.parity_stride + (unsigned)(((pt * 8 + NJ * w + j) * NKH + s) * 1024);
  };

  // ---- state
  v8bfw ones;
  {
    const __bf16 o1 = (__bf16)(hb == 0 ? 1.0f : 0.0f);
    ones = v8bfw{o1, o1, o1, 0, 0, 0, 0, 0};
  }
  v4f c[NP][TPM][NJ];
#pragma unroll
  for (int pt = 0; pt < NP; ++pt)
#pragma unroll
    for (int r = 0; r < TPM; ++r)
#pragma unroll
      for (int j = 0; j < NJ; ++j) c[pt][r][j] = v4f{0.f, 0.f, 0.f, 0.f};
  v16f acc[NP][TPM][NJ];          // set pt: accumulators of part pt (the other set is in its gate phase)
  v8bfw win[WS][NJ];
  v4f sk0[NJ], sk1[NJ];
  unsigned tagv = 0;

  const char* const lds_a = smem + lane * 16;
  auto arec = [&](int r, int k) { return __builtin_bit_cast(v8bfw, *reinterpret_cast<const v4f*>(lds_a + (r * KT + k) * 1024)); };

  auto load_group = [&](auto tc, int pt, unsigned tt, int par) {
    constexpr int T = decltype(tc)::value;
    static_for<4>([&](auto i) {
      constexpr int I = decltype(i)::value;
      static_for<NJ>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if constexpr (T < GX) {
          if constexpr (!(ABL & 16)) win[4 * (T % WG) + I][J] = __builtin_bit_cast(v8bfw, bld4(rx0, vo0[pt][J], tt * st0 + 32 * (4 * T + I)));
        } else {
          if constexpr (!(ABL & 8)) win[4 * (T % WG) + I][J] = __builtin_bit_cast(v8bfw, bld4_l2(rhx, vlane, hx_off(par, pt, J, 4 * (T - GX) + I)));
        }
      });
    });
  };
  auto load_skip = [&](int pt, unsigned tt) {
    static_for<NJ>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      sk0[J] = bld4(rx2, vo2[pt][J], tt * st2);
      sk1[J] = bld4(rx2, vo2[pt][J], tt * st2 + 16);
    });
  };
  auto load_tags = [&](int pt) {
    tagv = __hip_atomic_load(tag_base + (pt * 8 + w) * 8 + (lane % CL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto tags_ready = [&](unsigned need) { return __builtin_amdgcn_ballot_w64(tagv < need) == 0; };
  auto wait_tags = [&](int pt, unsigned need) {
    if (tags_ready(need)) return;
    for (unsigned spins = 0;; ++spins) {
      __builtin_amdgcn_s_sleep(16);
      load_tags(pt);
      if (tags_ready(need)) return;
      if (spins > kClusterSpinLimit) {
        if (lane == 0) __hip_atomic_store(cp.status, 0x20000u | (unsigned)(cg & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_trap();
      }
    }
  };
  const rsrc_t rtag_0 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (0 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  const rsrc_t rtag_1 = __builtin_amdgcn_make_buffer_rsrc(tag_base + (1 * 8 + w) * 8 + m, 0, 4, 0x00020000);
  unsigned pub_dep = 0;

  // ---- the epilogue of a part, in slots: slot s < KG = chunk (s % 3) of tile (s / 3): tile t = (j = t / TPM, r = t % TPM);
  // after a tile group's last chunk its outputs and operand records are stored; slot KPUB stores the tag.
  v4f g_i, g_f, g_g, g_o;                                // gates of the tile in flight
  v4bfw hq[TPM];                                         // h_t of the current sequence tile's four gate-row tiles
  auto epilogue_slot = [&](auto sc, auto ppc, int pstep) {
    constexpr int S = decltype(sc)::value, PP = decltype(ppc)::value;
    if constexpr (S < KG) {
      constexpr int T = S / 3, CH = S % 3, J = T / TPM, R = T % TPM;
      const v16f& ac = acc[PP][R][J];
      if constexpr (CH == 0) {
        if constexpr (ABL & 2) {
          g_i = v4f{ac[0], ac[1], ac[2], ac[3]};
          g_f = v4f{ac[4], ac[5], ac[6], ac[7]};
        } else {
          g_i = sigmoid4(v4f{ac[0], ac[1], ac[2], ac[3]});
          g_f = sigmoid4(v4f{ac[4], ac[5], ac[6], ac[7]});
        }
      } else if constexpr (CH == 1) {
        if constexpr (ABL & 2) {
          g_g = v4f{ac[8], ac[9], ac[10], ac[11]};
          g_o = v4f{ac[12], ac[13], ac[14], ac[15]};
        } else {
          g_g = tanh4(v4f{ac[8], ac[9], ac[10], ac[11]});
          g_o = sigmoid4(v4f{ac[12], ac[13], ac[14], ac[15]});
        }
      } else {
        v4f cn, hn;
        if constexpr (ABL & 2) {
          cn = c[PP][R][J] * 0.5f + g_i + g_g;
          hn = cn * 0.5f + g_f + g_o;
        } else {
          cn = cell4(g_f, c[PP][R][J], g_i, g_g);
          hn = mul_rn4(g_o, tanh4(cn));
        }
        c[PP][R][J] = cn;
        hq[R] = __builtin_convertvector(hn, v4bfw);
        if constexpr (R == TPM - 1) {                    // the sequence tile is complete: outputs, operand records
          const unsigned ptt = rev ? p.nsteps - 1 - pstep : pstep;
          const unsigned oo = ptt * sto;
          if constexpr (!(ABL & 32)) {
            const v2u d0 = __builtin_bit_cast(v2u, hq[0]), d1 = __builtin_bit_cast(v2u, hq[1]);
            const v2u d2 = __builtin_bit_cast(v2u, hq[2]), d3 = __builtin_bit_cast(v2u, hq[3]);
            const auto s00 = __builtin_amdgcn_permlane32_swap(d0[0], d2[0], false, false);
            const auto s01 = __builtin_amdgcn_permlane32_swap(d0[1], d2[1], false, false);
            const auto s10 = __builtin_amdgcn_permlane32_swap(d1[0], d3[0], false, false);
            const auto s11 = __builtin_amdgcn_permlane32_swap(d1[1], d3[1], false, false);
            const v4u lo = {s00[0], s01[0], s00[1], s01[1]};
            const v4u hi = {s10[0], s11[0], s10[1], s11[1]};
            if (valid[PP][J]) {
              __builtin_amdgcn_raw_buffer_store_b128(lo, ro, voo[PP][J], oo, 0);
              __builtin_amdgcn_raw_buffer_store_b128(hi, ro, voo[PP][J], oo + 16, 0);
            }
          }
          if constexpr (!(ABL & 64)) {
            const int wpar = pstep & 1;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, join8(hq[0], hq[1])), rhx, vlane, hx_off(wpar, PP, J, 2 * m), 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, join8(hq[2], hq[3])), rhx, vlane, hx_off(wpar, PP, J, 2 * m + 1), 16);
            if constexpr (J == NJ - 1) {                 // behind the wave's last record store: the load the tag store waits for
              asm volatile("" ::: "memory");
              pub_dep = __builtin_bit_cast(unsigned, bld1(rw, 0, 0));
              asm volatile("" ::: "memory");
            }
          }
        }
      }
    } else if constexpr (S == KPUB) {
      if constexpr (!(ABL & 64)) {
        unsigned tval = (unsigned)pstep + 1;
        asm volatile("; tag store ordered behind %1" : "+v"(tval) : "v"(pub_dep));
        __builtin_amdgcn_raw_buffer_store_b32(tval, PP == 0 ? rtag_0 : rtag_1, lane * 4, 0, 16);
      }
    }
  };

  // ---- prologue
  const unsigned tt_first = rev ? p.nsteps - 1 : 0;
  load_group(ic<0>{}, 0, tt_first, 0);
  load_skip(0, tt_first);
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // FIRST: the very first part of the launch has no previous part whose epilogue it could carry
  auto part_step = [&](auto ptc, int step, auto firstc) {
    constexpr int PT = decltype(ptc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr int PN = (PT + 1) % NP, PP = (PT + NP - 1) % NP;
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const int nstep = PT + 1 < NP ? step : (step + 1 < p.nsteps ? step + 1 : step);
    const unsigned ttn = rev ? p.nsteps - 1 - nstep : nstep;
    const int par = (step + 1) & 1;
    const int pstep = PT > 0 ? step : step - 1;          // the step the previous part belongs to

    v8bfw a[TPM];
    static_for<TPM>([&](auto r) { a[decltype(r)::value] = arec(decltype(r)::value, 0); });
    static_for<KT>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      constexpr int POS = (K >= 1 && K <= NB0) ? K - 1 : (K >= NB0 + 2 ? K - 2 : -1);
      if constexpr (POS >= 0 && POS % 4 == 0) {          // a window group starts: request the next one
        constexpr int G = POS / 4, T = G + 1;
        if constexpr (G == GTAG) {
          if constexpr (!(ABL & 1)) {
            if (step > 0) wait_tags(PT, (unsigned)step);
          }
        }
        if constexpr (T < GP)
          load_group(ic<T>{}, PT, tt, par);
        else
          load_group(ic<T - GP>{}, PN, ttn, 0);
        if constexpr (G == GX) load_skip(PN, ttn);
      }
      if constexpr (K == KG + 2) load_tags(PN);          // looked at GTAG groups into the next part
      // ---- A operands one K-step ahead; 8 MFMAs with the previous part's epilogue slot between them
      v8bfw an[TPM];
      if constexpr (K + 1 < KT) static_for<TPM>([&](auto r) { an[decltype(r)::value] = arec(decltype(r)::value, K + 1); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<NJ>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        v8bfw bop;
        if constexpr (K == 0)
          bop = ones;
        else if constexpr (K == NB0 + 1)
          bop = join8(__builtin_convertvector(sk0[J], v4bfw), __builtin_convertvector(sk1[J], v4bfw));
        else
          bop = win[POS % WS][J];
        static_for<TPM>([&](auto r) {
          constexpr int R = decltype(r)::value;
          if constexpr (K == 0) {
            const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[PT][R][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[R], bop, z, 0, 0, 0);
          } else {
            acc[PT][R][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[R], bop, acc[PT][R][J], 0, 0, 0);
          }
        });
      });
      if constexpr (!FIRST) epilogue_slot(kc, ic<PP>{}, pstep);
      // one MFMA, then a handful of the slot's vector instructions, eight times: the slot's work is issued in the 28 cycles
      // an MFMA leaves free behind its own issue
      if constexpr (ABL & 256) {
        static_for<8>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (K + 1 < KT) static_for<TPM>([&](auto r) { a[decltype(r)::value] = an[decltype(r)::value]; });
    });
  };

  part_step(ic<0>{}, 0, ic<1>{});
  part_step(ic<1>{}, 0, ic<0>{});
  for (int step = 1; step < p.nsteps; ++step) static_for<NP>([&](auto pt) { part_step(pt, step, ic<0>{}); });
  // the last part's epilogue has no matrix phase to hide under
  static_for<KPUB + 1>([&](auto s) { epilogue_slot(s, ic<NP - 1>{}, p.nsteps - 1); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int H, int NB0, int NB2, int FLAGS, int ABL = 0>
int launch_bf16d_k(const LstmParams& p, const ClusterParams& cp, hipStream_t st) {
  constexpr int KT = 1 + NB0 + NB2 + H / 16, CL = cluster_members(H);
  const size_t lds = (size_t)(H / 8 / CL) * KT * 1024;
  auto k = lstm_bf16d_kernel<H, NB0, NB2, FLAGS, ABL>;
  FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nwg = 8 * CL * ((cp.ncl + 7) / 8);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, st, p, cp);
  FNSSL_CHECK_LAUNCH("lstm_bf16d_kernel");
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
