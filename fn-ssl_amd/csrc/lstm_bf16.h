// bf16-MFMA variant of the LSTM recurrence (BASELINE config 3: "IPDnet ... bf16 on 1 MI355X").
//
// Same formulation and the same tensors as the fp32 kernels (activations and the cell state stay fp32 in
// memory, gates are evaluated in fp32, accumulation is fp32); what changes is the arithmetic of the matrix
// product: weights are stored in the stream as bf16, the [x_t | h_{t-1}] operands are rounded to bf16 when
// they enter the MFMA, and one v_mfma_f32_16x16x32_bf16 replaces eight v_mfma_f32_16x16x4_f32
// (16x the matrix rate).  A "quad" of the stream is again 4 records of 1 KiB = the A operands of the four
// gate MFMAs, but now for a PAIR of 16-channel blocks (32 channels): lane (i, g) holds, for unit 16s + i,
// the 8 weights of channels 16*b0 + 4g + {0..3} and 16*b1 + 4g + {0..3} — i.e. exactly the two float4 the
// lane loads for blocks b0 and b1 (K is permuted consistently, as in the fp32 kernel).
//
// With the MFMAs 16x cheaper, everything else is the kernel: this version is written for the low-occupancy
// launches IPDnet produces (one or two waves per SIMD, registers are plentiful) —
//   * x_t is loaded ONCE per step (not once per hidden slice), converted once, and held as packed bf16;
//     the next step's blocks are requested during the current step;
//   * h_{t-1} and the cell state never leave the registers: each slice's D fragment is packed to bf16 into
//     the operand of the next step (double-buffered), h is only STORED for the next layer;
//   * the structure of a slice is compile-time (immediate LDS offsets, ring commits at fixed positions) like
//     lstm_static.h.
// Numerics: bf16 operand rounding (8-bit mantissa) — not bit-comparable with the fp32 path; the oracle
// restates exactly this rounding (oracle/fnssl_oracle.py, bf16=True) and the tests state the tolerance.
#pragma once

#include "lstm_static.h"

namespace fnssl_lstm {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v8bf pack_bf16(v4f a, v4f b) {
  const v4bf lo = __builtin_convertvector(a, v4bf), hi = __builtin_convertvector(b, v4bf);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// quads per slice: bias + pairs of src0 blocks + pairs of src2 blocks + pairs of hidden slices
__host__ __device__ inline int bf16_quads_per_slice(int c0, int c2, int H) {
  return 1 + ((c0 >> 4) + 1) / 2 + ((c2 >> 4) + 1) / 2 + (H >> 5);
}

// NV0 / NV2: 16-channel blocks of the summed / concatenated input (c0 = 16 NV0, c2 = 16 NV2)
template <int H, int NW, int M, int NV0, int NV2, int CHQ, int PAD>
__global__ void __launch_bounds__(NW * 64) lstm_bf16_kernel(const LstmParams p) {
  constexpr int NS = H / 16, NHP = NS / 2;
  constexpr int P0 = (NV0 + 1) / 2, P2 = (NV2 + 1) / 2, NXP = P0 + P2;
  constexpr int QPS = 1 + NXP + NHP, VQ = QPS + PAD, CH = 4 * CHQ;
  static_assert(VQ % CHQ == 0 && CH <= NW * M && NS % 2 == 0, "geometry");
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

  unsigned vo0 = 0, vo2 = 0, voo = 0;
  const rsrc_t rx0 = NV0 ? split_addr(p.src0.p, qo * p.src0.so + qi * p.src0.si, 4 * g, vo0) : make_rsrc(p.out);
  const rsrc_t rx2 = NV2 ? split_addr(p.src2.p, qo * p.src2.so + qi * p.src2.si, 4 * g, vo2) : rx0;
  const rsrc_t ro = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo);
  const rsrc_t rw = make_rsrc(p.wpack[dir]);
  const unsigned st0 = NV0 ? (unsigned)(p.src0.st * 4) : 0u, st2 = NV2 ? (unsigned)(p.src2.st * 4) : 0u;
  const unsigned sto = (unsigned)(p.out_st * 4);
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;

  // ---- weight ring (as in lstm_static.h) -----------------------------------------------------
  char* const lds_rd = smem + lane * 16;
  char* const lds_wr = smem + w * 1024 + lane * 16;
  int wslot = 0, rslot = 0, src_rec = 0, src_vq = 0;
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
      src_rec -= PAD * 4;
      if (src_rec == NS * QPS * 4) src_rec = 0;
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (w + m * NW < CH) *reinterpret_cast<v4f*>(lds_wr + wslot * (CH * 1024) + m * (NW * 1024)) = stg[m];
    wslot ^= 1;
  };
  auto sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  issue_loads();
  stage_write();
  sync();
  issue_loads();
  const char* cb = lds_rd;
  auto rec = [&](auto ql, int j) {
    return *reinterpret_cast<const v4f*>(cb + decltype(ql)::value * 4096 + j * 1024);
  };
  // A operands: the four records of a quad are requested one whole quad ahead (the four MFMAs of a quad take
  // only ~70 cycles, less than an LDS round trip, so a half-quad lookahead as in the fp32 kernels stalls on
  // every quad); an[0..3] always hold the NEXT quad to be consumed.
  v4f an0 = rec(ic<0>{}, 0), an1 = rec(ic<0>{}, 1), an2 = rec(ic<0>{}, 2), an3 = rec(ic<0>{}, 3);
  v4f acc[4];
  // after quad QI: request quad QI + 1 (same chunk), or commit / re-read at a chunk end
  auto ring_next = [&](auto qi_c) {
    constexpr int QL = decltype(qi_c)::value % CHQ;
    if constexpr (QL + 1 == (CHQ + 1) / 2 && CHQ > 1) stage_write();
    if constexpr (QL + 1 == CHQ) {
      if constexpr (CHQ == 1) stage_write();
      sync();
      issue_loads();
      rslot ^= 1;
      cb = lds_rd + rslot * (CH * 1024);
      an0 = rec(ic<0>{}, 0);
      an1 = rec(ic<0>{}, 1);
      an2 = rec(ic<0>{}, 2);
      an3 = rec(ic<0>{}, 3);
    }
  };
#define MFMA_BF(ACC, AV, BV) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, AV), BV, ACC, 0, 0, 0)
  // one quad: the four gate MFMAs of a 32-channel pair; B is the packed operand
#define BQUAD(QI, BV)                                                              \
  do {                                                                             \
    const v4f c0_ = an0, c1_ = an1, c2_ = an2, c3_ = an3;                          \
    if constexpr (((QI) % CHQ) + 1 < CHQ) {                                        \
      an0 = rec(ic<((QI) % CHQ) + 1>{}, 0);                                        \
      an1 = rec(ic<((QI) % CHQ) + 1>{}, 1);                                        \
      an2 = rec(ic<((QI) % CHQ) + 1>{}, 2);                                        \
      an3 = rec(ic<((QI) % CHQ) + 1>{}, 3);                                        \
    }                                                                              \
    MFMA_BF(acc[0], c0_, BV);                                                      \
    MFMA_BF(acc[1], c1_, BV);                                                      \
    MFMA_BF(acc[2], c2_, BV);                                                      \
    MFMA_BF(acc[3], c3_, BV);                                                      \
    ring_next(ic<(QI)>{});                                                         \
  } while (0)

  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  // x_t as packed bf16 pairs (held for the whole step) and the raw fp32 blocks of the next step
  v8bf xb[NXP > 0 ? NXP : 1];
  v4f xraw[2 * NXP > 0 ? 2 * NXP : 1];
  auto request_x = [&](unsigned tt) {
    static_for<P0>([&](auto pi) {
      constexpr int PI = decltype(pi)::value;
      xraw[2 * PI] = bld4(rx0, vo0, tt * st0 + 128 * PI);
      xraw[2 * PI + 1] = (2 * PI + 1 < NV0) ? bld4(rx0, vo0, tt * st0 + 128 * PI + 64) : zero4;
    });
    static_for<P2>([&](auto pi) {
      constexpr int PI = decltype(pi)::value;
      xraw[2 * (P0 + PI)] = bld4(rx2, vo2, tt * st2 + 128 * PI);
      xraw[2 * (P0 + PI) + 1] = (2 * PI + 1 < NV2) ? bld4(rx2, vo2, tt * st2 + 128 * PI + 64) : zero4;
    });
  };
  // h_{t-1} as packed bf16 pairs of hidden slices; hnew collects this step's h
  v8bf hcur[NHP], hnew[NHP];
#pragma unroll
  for (int i = 0; i < NHP; ++i) hcur[i] = pack_bf16(zero4, zero4);
  request_x(rev ? p.nsteps - 1 : 0);
  // The cell state lives in registers too.  (A variant that kept it in the lane-private scratch, like the fp32
  // kernels, read back wrong values in lanes 12-15 of even slices on ROCm 7.2 / gfx950 — addresses that
  // coincide with h-store offsets, i.e. a buffer-descriptor mix-up at the 106-SGPR limit, independent of
  // waitcnts and cache policy; registers are faster anyway.)
  v4f creg[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) creg[i] = zero4;

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;
    const unsigned oo = tt * sto;
    static_for<NXP>([&](auto pi) { xb[pi.value] = pack_bf16(xraw[2 * pi.value], xraw[2 * pi.value + 1]); });
    request_x(ttn);   // lands during this step

    static_for<NS>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      const v4f cprev = creg[S];
      acc[0] = an0;   // bias quad (fp32)
      acc[1] = an1;
      acc[2] = an2;
      acc[3] = an3;
      if constexpr (1 < CHQ) {
        an0 = rec(ic<1>{}, 0);
        an1 = rec(ic<1>{}, 1);
        an2 = rec(ic<1>{}, 2);
        an3 = rec(ic<1>{}, 3);
      }
      ring_next(ic<0>{});
      static_for<NXP>([&](auto pi) { BQUAD(1 + decltype(pi)::value, xb[decltype(pi)::value]); });
      static_for<NHP>([&](auto hp) { BQUAD(1 + NXP + decltype(hp)::value, hcur[decltype(hp)::value]); });
      static_for<PAD>([&](auto u) {
        constexpr int QP = QPS + decltype(u)::value;
        if constexpr ((QP % CHQ) + 1 < CHQ) {
          an0 = rec(ic<(QP % CHQ) + 1>{}, 0);
          an1 = rec(ic<(QP % CHQ) + 1>{}, 1);
          an2 = rec(ic<(QP % CHQ) + 1>{}, 2);
          an3 = rec(ic<(QP % CHQ) + 1>{}, 3);
        }
        ring_next(ic<QP>{});
      });
      const v4f ig = sigmoid4(acc[0]);
      const v4f fg = sigmoid4(acc[1]);
      const v4f gg = tanh4(acc[2]);
      const v4f og = sigmoid4(acc[3]);
      const v4f cn = cell4(fg, cprev, ig, gg);
      const v4f hn = mul_rn4(og, tanh4(cn));
      creg[S] = cn;
      if (valid) bst4(hn, ro, voo, oo + 64 * S);
      // the next step's operand: slice S is the low (even S) or high (odd S) half of pair S/2
      const v4bf hb = __builtin_convertvector(hn, v4bf);
      if constexpr ((S & 1) == 0)
        hnew[S / 2] = __builtin_shufflevector(hb, hb, 0, 1, 2, 3, 0, 1, 2, 3);
      else
        hnew[S / 2] = __builtin_shufflevector(hnew[S / 2], __builtin_shufflevector(hb, hb, 0, 1, 2, 3, 0, 1, 2, 3), 0,
                                              1, 2, 3, 8, 9, 10, 11);
    });
#pragma unroll
    for (int i = 0; i < NHP; ++i) hcur[i] = hnew[i];
  }
#undef BQUAD
#undef MFMA_BF
}

template <int H, int NW, int M, int NV0, int NV2, int CHQ, int PAD>
int launch_bf16_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = (size_t)2 * CHQ * 4096;
  static_assert(2 * CHQ * 4096 <= 160 * 1024, "ring does not fit the LDS");
  auto k = lstm_bf16_kernel<H, NW, M, NV0, NV2, CHQ, PAD>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_bf16_kernel");
  return FNSSL_OK;
}

// kNoStatic when (H, NW, c0, c2) has no bf16 instantiation
int launch_bf16(const LstmParams& p, int H, int NW, int nwg, hipStream_t st);

}  // namespace fnssl_lstm
