// Training kernels of the LSTM layers: the forward that also saves the gate
// activations, and back-propagation through time.  Host entry points: lstm_train.hip.
//
// Backward formulation.  With the forward's transposed product G^T = W [x | h]^T, the gradient of the
// layer input and of the previous hidden state is again a transposed product
//     [dx_t | dh_{t-1}]^T [(C0g + H) x 16] = [W_ih | W_hh]^T [(C0g + H) x 4H] * da_t^T [4H x 16]
// with the SAME operand roles: A = a 16-row tile of the (transposed) weights streamed through the LDS
// ring, B = the pre-activation gradients da_t with lane <-> sequence.  A wave owns 16 sequences, walks
// the time axis against the forward direction and alternates two phases per step:
//   A  (VALU)  per 16-unit hidden slice: read i, f, g, o, c_t, c_{t-1} (forward reserve), the upstream
//              gradient and the carried dh / dc, form da_{i,f,g,o}, write them to the dA tensor (the
//              weight-gradient GEMMs consume it afterwards) and carry dc;
//   B  (MFMA)  stream [W_ih | W_hh]^T: for every group of 4 x 16 output channels accumulate over the 4H
//              gate units (B operand re-read from the dA rows this wave has just written), then store
//              dx_t (input-gradient tensor) or carry dh_{t-1} (lane-private scratch).
// The output channels are arranged like a forward "fake LSTM" with 4 "gates" = the four quarters of the
// padded output range, so the packed stream has the forward's record structure (bias quad = zeros) and
// the D fragments land on the lanes that own those units in phase A of the next step.
#pragma once

#include "lstm_kernel.h"
#include "tuning.h"

namespace fnssl_lstm {

struct BwdParams {
  const float* reserve;
  View dh;                     // upstream gradient wrt the layer output (channel offset dir * H)
  float* da;                   // [seq, step, ndir * 4H] view (written)
  long long da_so, da_si, da_st;
  float* dx;                   // [seq, step, ndir * c0g] view (written), null when c0g == 0
  long long dx_so, dx_si, dx_st;
  const float* wpack[2];
  float* scratch;              // per wave 2 * NS records: carried dh, carried dc
  int c0g, co_pad;             // input channels that need a gradient; padded output range (multiple of 64)
  int nseq, q_inner, nsteps, ndir, ntasks;
  int wgs_per_dir, task0, task1;
  int quads_per_slice, chq, pad;
  // guarded fallback launch behind lstm_bwdc_kernel (lstm_bwdc.h): runs only if *guard != 0 (LstmParams.guard)
  const unsigned* guard = nullptr;
  unsigned* fallback_count = nullptr;   // optional device counter: block 0 of a guarded launch that does run counts the call
  int dry = 0;                 // host only (fnssl_lstm_backward_plan): no launch
};

inline int bwd_co_pad(int c0g, int H) { return (c0g + H + 63) / 64 * 64; }
inline int bwd_quads_per_slice(int H) { return 1 + (4 * H) / 16; }

// SPLIT > 1: SPLIT waves share one 16-sequence group (see lstm_rec_kernel): each does its share of the hidden
// slices in phase A and of the output slices in phase B; dA rows and the carried dh are exchanged through
// memory with a workgroup barrier after each phase.
// DIRECT (split launches): no LDS ring and no ring barriers — with few waves on the chip the L2 has bandwidth
// to spare, so each wave streams its own share of the weight quads straight from L2 into a 4-quad-deep
// register pipeline (16 records in flight); only the two per-step barriers of the split remain.
template <int H, int NW, int M, int SPLIT = 1, bool DIRECT = false>
__global__ void __launch_bounds__(NW * 64) lstm_bwd_kernel(const BwdParams p) {
  constexpr int NS = H / 16;
  constexpr int NSL = NS / SPLIT;
  constexpr int NVB = 4 * H / 16;   // 16-channel blocks of one dA row
  static_assert(NS % SPLIT == 0 && NW % SPLIT == 0, "split geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.guard && __hip_atomic_load(p.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // uniform for the grid
  if (p.guard && p.fallback_count && blockIdx.x == 0 && threadIdx.x == 0 && p.task0 == 0) atomicAdd(p.fallback_count, 1u);
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int part = SPLIT > 1 ? w % SPLIT : 0;
  const int task = p.task0 + wg * (NW / SPLIT) + w / SPLIT;
  const bool tvalid = task < p.task1;
  int q = task * 16 + n;
  const bool valid = q < p.nseq && tvalid;
  if (q >= p.nseq) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  unsigned vdh = 0, vda = 0, vdx = 0;
  const rsrc_t rdh = split_addr(p.dh.p, qo * p.dh.so + qi * p.dh.si, dir * H + 4 * g, vdh);
  const rsrc_t rda = split_addr(p.da, qo * p.da_so + qi * p.da_si, dir * 4 * H + 4 * g, vda);
  const rsrc_t rdx = p.c0g ? split_addr(p.dx, qo * p.dx_so + qi * p.dx_si, dir * p.c0g + 4 * g, vdx) : rdh;
  const rsrc_t rres = make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                                ((size_t)dir * p.ntasks + (tvalid ? task : 0)) * p.nsteps *
                                    (size_t)(NS * kReserveRecs * 1024));
  // carried dh / dc: one region per (direction, group), shared by the waves of a split group
  const rsrc_t rsc = make_rsrc(reinterpret_cast<const char*>(p.scratch) +
                               ((size_t)dir * (p.ntasks + 16) + (tvalid ? task : p.ntasks + w)) * (2 * NS * 1024));
  const unsigned sdh = (unsigned)(p.dh.st * 4), sda = (unsigned)(p.da_st * 4), sdx = (unsigned)(p.dx_st * 4);
  const unsigned vlane = lane * 16;
  const bool rev = dir == 1;
  const int nsol = (p.co_pad >> 6) / SPLIT;   // output slices (4 x 16 channels each) of this wave
  const int hq = p.co_pad >> 2;               // channels per output quarter

  WStream<NW, M, 1> ws;
  ws.nobar = false;
  v4f a0, a1;
  if (!DIRECT) {
    ws.init(p.wpack[dir], lane, w, p.quads_per_slice, nsol, p.chq, p.pad, smem, SPLIT, part);
    a0 = ws.record(0);
    a1 = ws.record(1);
  }
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f acc[4];
  // DIRECT: this wave's data quads (the zero "bias" quad of each output slice is skipped) form a cyclic
  // sequence of nsol * NVB quads; ar[k] holds quad (cursor + k) of it
  const rsrc_t rwd = make_rsrc(p.wpack[dir]);
  const int dq_total = nsol * NVB;
  int dq_next = 0;                       // next data quad to request
  auto dq_load = [&](v4f* dst) {
    const int sl = dq_next / NVB, qq = dq_next - sl * NVB;
    const unsigned rec0 = (unsigned)(((part * nsol + sl) * p.quads_per_slice + 1 + qq) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = bld4(rwd, vlane, (rec0 + j) * 1024u);
    dq_next = dq_next + 1 == dq_total ? 0 : dq_next + 1;
  };
  v4f ar0[4], ar1[4], ar2[4], ar3[4];
  if (DIRECT) {
    dq_load(ar0);
    dq_load(ar1);
    dq_load(ar2);
    dq_load(ar3);
  }
#define DQUAD(AR, B0, B1, B2, B3) \
  do {                            \
    MFMA4(acc, AR[0], B0);        \
    MFMA4(acc, AR[1], B1);        \
    MFMA4(acc, AR[2], B2);        \
    MFMA4(acc, AR[3], B3);        \
    dq_load(AR);                  \
  } while (0)

#define QUAD(B0, B1, B2, B3)                          \
  do {                                                \
    const v4f a2_ = ws.record(2), a3_ = ws.record(3); \
    __builtin_amdgcn_sched_barrier(0);                \
    MFMA4(acc, a0, B0);                               \
    MFMA4(acc, a1, B1);                               \
    ws.peek_next(a0, a1);                             \
    __builtin_amdgcn_sched_barrier(0);                \
    MFMA4(acc, a2_, B2);                              \
    MFMA4(acc, a3_, B3);                              \
    if (ws.advance()) {                               \
      a0 = ws.record(0);                              \
      a1 = ws.record(1);                              \
    }                                                 \
  } while (0)

  for (int step = 0; step < p.nsteps; ++step) {
    // the forward direction ran t = 0 .. T-1, so its gradient flows T-1 .. 0; the reverse one the other way
    const unsigned tt = rev ? step : p.nsteps - 1 - step;
    const bool has_prev = step + 1 < p.nsteps;            // the forward pass had a step before tt
    const unsigned tp = has_prev ? (rev ? tt + 1 : tt - 1) : tt;
    const unsigned oa = tt * sda;

    // ---- phase A: gate gradients of every hidden slice ---------------------------------------
    // the 7-9 operand loads of slice sl + 1 are requested before the arithmetic and the 5 stores of slice sl (rolled
    // loop: the compiler does not move them across the back edge, and every slice started with a memory round trip)
    struct SliceIn {
      v4f ig, fg, gg, og, ct, cp, dh, dhc, dc;
    };
    auto load_slice = [&](int sl, SliceIn& in) {
      const int s = SPLIT > 1 ? part * NSL + sl : sl;
      const unsigned rb = (tt * NS + s) * (kReserveRecs * 1024);
      in.ig = bld4(rres, vlane, rb);
      in.fg = bld4(rres, vlane, rb + 1024);
      in.gg = bld4(rres, vlane, rb + 2048);
      in.og = bld4(rres, vlane, rb + 3072);
      in.ct = bld4(rres, vlane, rb + 4096);
      in.dh = bld4(rdh, vdh, tt * sdh + 64 * s);
      in.cp = zero4;
      in.dhc = zero4;
      in.dc = zero4;
      if (has_prev) in.cp = bld4(rres, vlane, (tp * NS + s) * (kReserveRecs * 1024) + 4096);
      if (step > 0) {
        in.dhc = bld4(rsc, vlane, s * 1024);
        in.dc = bld4(rsc, vlane, (NS + s) * 1024);
      }
    };
    SliceIn nxt;
    load_slice(0, nxt);
    for (int sl = 0; sl < NSL; ++sl) {
      const int s = SPLIT > 1 ? part * NSL + sl : sl;
      const SliceIn in = nxt;
      if (sl + 1 < NSL) load_slice(sl + 1, nxt);
      const v4f ig = in.ig, fg = in.fg, gg = in.gg, og = in.og, ct = in.ct, cp = in.cp;
      v4f dh = in.dh + in.dhc, dc = in.dc;
      const v4f tc = tanh4(ct);
      const v4f one = v4f{1.f, 1.f, 1.f, 1.f};
      dc += dh * og * (one - tc * tc);
      const v4f dao = dh * tc * og * (one - og);
      const v4f dai = dc * gg * ig * (one - ig);
      const v4f daf = dc * cp * fg * (one - fg);
      const v4f dag = dc * ig * (one - gg * gg);
      bst4(dc * fg, rsc, vlane, (NS + s) * 1024);
      if (valid) {
        bst4(dai, rda, vda, oa + 64 * s);
        bst4(daf, rda, vda, oa + 4 * H + 64 * s);
        bst4(dag, rda, vda, oa + 8 * H + 64 * s);
        bst4(dao, rda, vda, oa + 12 * H + 64 * s);
      }
    }
    // the B operands below are the dA rows just written (by this wave, or by all waves of a split group):
    // wait until the stores have been performed
    if (SPLIT > 1)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- phase B: [dx | dh_prev]^T = [W_ih | W_hh]^T da^T -------------------------------------
    v4f x0 = bld4(rda, vda, oa), x1 = bld4(rda, vda, oa + 64), x2 = bld4(rda, vda, oa + 128),
        x3 = bld4(rda, vda, oa + 192);
    for (int sol = 0; sol < nsol; ++sol) {
      const int so = SPLIT > 1 ? part * nsol + sol : sol;
      if (DIRECT) {
        acc[0] = acc[1] = acc[2] = acc[3] = zero4;
#pragma unroll 1
        for (int v = 0; v < NVB; v += 4) {
          const unsigned nx = oa + 64 * ((v + 4) & (NVB - 1));
          DQUAD(ar0, x0.x, x0.y, x0.z, x0.w);
          x0 = bld4(rda, vda, nx);
          DQUAD(ar1, x1.x, x1.y, x1.z, x1.w);
          x1 = bld4(rda, vda, nx + 64);
          DQUAD(ar2, x2.x, x2.y, x2.z, x2.w);
          x2 = bld4(rda, vda, nx + 128);
          DQUAD(ar3, x3.x, x3.y, x3.z, x3.w);
          x3 = bld4(rda, vda, nx + 192);
        }
      } else {
        acc[0] = a0;   // "bias" quad of the stream: zeros
        acc[1] = a1;
        acc[2] = ws.record(2);
        acc[3] = ws.record(3);
        ws.peek_next(a0, a1);
        if (ws.advance()) {
          a0 = ws.record(0);
          a1 = ws.record(1);
        }
#pragma unroll 1
        for (int v = 0; v < NVB; v += 4) {
          // 4-deep operand ring; the blocks wrap around into the next output slice (same dA row)
          const unsigned nx = oa + 64 * ((v + 4) & (NVB - 1));
          QUAD(x0.x, x0.y, x0.z, x0.w);
          x0 = bld4(rda, vda, nx);
          QUAD(x1.x, x1.y, x1.z, x1.w);
          x1 = bld4(rda, vda, nx + 64);
          QUAD(x2.x, x2.y, x2.z, x2.w);
          x2 = bld4(rda, vda, nx + 128);
          QUAD(x3.x, x3.y, x3.z, x3.w);
          x3 = bld4(rda, vda, nx + 192);
        }
        for (int u = 0; u < p.pad; ++u) {
          ws.peek_next(a0, a1);
          if (ws.advance()) {
            a0 = ws.record(0);
            a1 = ws.record(1);
          }
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int ob = qq * hq + 16 * so;                 // first channel of this 16-channel block
        if (ob < p.c0g) {
          if (valid) bst4(acc[qq], rdx, vdx, tt * sdx + 4 * ob);
        } else if (ob < p.c0g + H) {
          bst4(acc[qq], rsc, vlane, ((ob - p.c0g) >> 4) * 1024);
        }
      }
    }
    if (SPLIT > 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // carried dh visible to the group
  }
#undef QUAD
#undef DQUAD
}

template <int H, int NW, int M, int SPLIT = 1, bool DIRECT = false>
int launch_bwd_k(const BwdParams& p, int nwg, hipStream_t st) {
  const size_t lds = DIRECT ? 0 : (size_t)2 * p.chq * SPLIT * 4096;
  auto k = lstm_bwd_kernel<H, NW, M, SPLIT, DIRECT>;
  if (p.dry) return FNSSL_OK;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_bwd_kernel");
  return FNSSL_OK;
}

template <int H>
int launch_bwd(int nw, int split, const BwdParams& p, int nwg, hipStream_t st) {
  if (split == 1) {
    switch (nw) {
      case 4: return launch_bwd_k<H, 4, 4>(p, nwg, st);
      case 8: return launch_bwd_k<H, 8, 4>(p, nwg, st);
      case 12: return launch_bwd_k<H, 12, 4>(p, nwg, st);
    }
  } else if (split == 4 && !fnssl::tune(FNSSL_TUNE_BWD_RING)) {
    // 4 waves per group: weights straight from L2 (narrow-band BPTT 31.8 -> 26.0 ms at config 4; with 2 waves
    // per group the LDS ring is as fast, r01 f_train_layers)
    if (nw == 4) return launch_bwd_k<H, 4, 1, 4, true>(p, nwg, st);
    if (nw == 8) return launch_bwd_k<H, 8, 1, 4, true>(p, nwg, st);
  } else if (split == 2 && nw == 4) {
    return launch_bwd_k<H, 4, 8, 2>(p, nwg, st);
  } else if (split == 4 && nw == 4) {
    return launch_bwd_k<H, 4, 8, 4>(p, nwg, st);
  } else if (split == 4 && nw == 8) {
    return launch_bwd_k<H, 8, 8, 4>(p, nwg, st);
  }
  fnssl::set_error("lstm_backward: unsupported geometry (%d waves, split %d)", nw, split);
  return FNSSL_E_INVALID;
}

// (nw, split): 4 / 8 / 12 waves per workgroup with one group per wave, or the split geometries for launches
// with fewer wave tasks than SIMDs: 2 or 4 waves per 16-sequence group (8 ring records staged per wave)
template <int H, int MODE>
int launch_save_m(int nw, int split, const LstmParams& p, int nwg, hipStream_t st) {
  if (split == 1) {
    switch (nw) {
      case 4: return launch_k<H, 4, 4, 1, MODE>(p, nwg, st);
      case 8: return launch_k<H, 8, 4, 1, MODE>(p, nwg, st);
      case 12: return launch_k<H, 12, 4, 1, MODE>(p, nwg, st);
    }
  } else if (split == 2 && nw == 4) {
    return launch_k<H, 4, 4, 1, MODE, false, 2>(p, nwg, st);
  } else if (split == 4 && nw == 4) {
    return launch_k<H, 4, 4, 1, MODE, false, 4>(p, nwg, st);
  } else if (split == 4 && nw == 8) {
    return launch_k<H, 8, 8, 1, MODE, false, 4>(p, nwg, st);
  }
  fnssl::set_error("lstm_forward (training): unsupported geometry (%d waves, split %d)", nw, split);
  return FNSSL_E_INVALID;
}

template <int H>
int launch_save(int nw, int split, const LstmParams& p, int mode, int nwg, hipStream_t st) {
  return (mode & kHas2) ? launch_save_m<H, kSave | kHas2>(nw, split, p, nwg, st)
                        : launch_save_m<H, kSave>(nw, split, p, nwg, st);
}

extern template int launch_bwd<128>(int, int, const BwdParams&, int, hipStream_t);
extern template int launch_bwd<256>(int, int, const BwdParams&, int, hipStream_t);
extern template int launch_save<128>(int, int, const LstmParams&, int, int, hipStream_t);
extern template int launch_save<256>(int, int, const LstmParams&, int, int, hipStream_t);

}  // namespace fnssl_lstm
