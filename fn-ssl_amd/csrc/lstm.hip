// LSTM recurrence for gfx950 (MI355X): replaces nn.LSTM on the FN-SSL hot path
// (reference FN-SSL/Model.py:25-29,38,46).
//
// Formulation.  One WAVE owns 16 sequences for the whole
// recurrence and computes the TRANSPOSED gate product with fp32 MFMA
//     G^T[4H x 16] = W[4H x K] * [x_t | h_{t-1}]^T [K x 16]
// using v_mfma_f32_16x16x4_f32: A = a 16-row weight tile, B = activations with
// lane <-> sequence.  The D fragment then has lane <-> sequence and registers <->
// hidden unit, which is exactly the B-operand layout of the next step, so h_t
// never leaves the register file between steps (K is permuted consistently in
// the packed weights).  Gates i,f,g,o of a 16-unit hidden "slice" are four
// accumulators of the same lane, so the cell update is pure per-lane VALU work.
//
// The weight stream (1 KiB "records" = one float4 per lane = the A operand of
// four MFMAs, one per gate) is identical for all waves of a workgroup.  Variant
// WMODE=1 stages it through a 2-slot LDS ring filled by plain (compiler-counted)
// global loads one chunk ahead; WMODE=0 reads it straight from L1/L2.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

struct View {
  const float* p;
  long long so, si, st;
};

struct LstmParams {
  View src0, src1, src2;
  float* out;
  long long out_so, out_si, out_st;
  const float* wpack[2];
  float* cscratch;
  int c0, c2;
  int nseq, q_inner, nsteps, ndir;
  int wgs_per_dir;
  int quads_per_slice;
  int chq, pad;   // ring chunk (quads) and per-slice padding (quads); 0 for direct variants
};

// ---- stream geometry (shared by packer and kernel) -------------------------
// per hidden slice (16 units):  quad 0           : 4 bias records (acc init, gate q)
//                               seg0 vec quads    : c0/16 quads, record j <-> k = 16v + 4g + j
//                               seg0 scalar quads : (c0%16)/4 quads, record 0 <-> k = base + g
//                               seg2 vec / scalar : same for the concatenated input
//                               h quads           : H/16 quads, record j <-> k = 16s' + 4g + j
__host__ __device__ inline int quads_per_slice(int c0, int c2, int H) {
  return 1 + (c0 >> 4) + ((c0 & 15) >> 2) + (c2 >> 4) + ((c2 & 15) >> 2) + (H >> 4);
}

__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanh_f(float x) {
  // 1 - 2/(e^{2x}+1): saturates cleanly at +-1, abs error ~1e-7
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}
__device__ __forceinline__ v4f sigmoid4(v4f a) {
  return v4f{sigmoid_f(a.x), sigmoid_f(a.y), sigmoid_f(a.z), sigmoid_f(a.w)};
}
__device__ __forceinline__ v4f tanh4(v4f a) {
  return v4f{tanh_f(a.x), tanh_f(a.y), tanh_f(a.z), tanh_f(a.w)};
}

#define MFMA4(ACC, AV, BV)                                                          \
  do {                                                                              \
    ACC[0] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).x, (BV), ACC[0], 0, 0, 0);   \
    ACC[1] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).y, (BV), ACC[1], 0, 0, 0);   \
    ACC[2] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).z, (BV), ACC[2], 0, 0, 0);   \
    ACC[3] = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).w, (BV), ACC[3], 0, 0, 0);   \
  } while (0)

// ---- addressing ------------------------------------------------------------
// Every global access is a raw buffer op: 64-bit wave-uniform base in an SGPR
// descriptor, one 32-bit per-lane byte offset VGPR per tensor, and the moving
// part (step, block, record) in the scalar offset.  The host checks that the
// per-descriptor extents fit in 32 bits.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ v4f bld4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float bld1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bst4(v4f d, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, d), r, voff, soff, 0);
}

// Split a per-lane address  base + off_floats (+ extra floats)  into a descriptor
// whose base is the wave's minimum and a per-lane byte offset >= 0.
__device__ __forceinline__ rsrc_t split_addr(const float* base, long long off_floats, int extra_floats,
                                             unsigned& voff) {
  long long mn = off_floats;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long o = __shfl_xor(mn, d, 64);
    mn = o < mn ? o : mn;
  }
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(mn & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)mn >> 32));
  const long long mnu = (long long)(((unsigned long long)hi << 32) | lo);
  voff = (unsigned)((off_floats - mnu) * 4) + (unsigned)(extra_floats * 4);
  return make_rsrc(base + mnu);
}

// Weight-stream reader.  WMODE 0: straight from global.  WMODE 1: 2-slot LDS ring,
// CH = NW*M records per chunk, every wave stages M records per chunk in registers.
template <int NW, int M, int WMODE>
struct WStream;

template <int NW, int M>
struct WStream<NW, M, 0> {
  rsrc_t rw;
  unsigned vlane;
  unsigned cur;   // byte offset of the current quad in the stream
  unsigned bytes_per_step;
  __device__ __forceinline__ void init(const float* wp, int lane, int /*w*/, int qps, int nslices, int /*chq*/,
                                       int /*pad*/, char* /*smem*/) {
    rw = make_rsrc(wp);
    vlane = lane * 16;
    cur = 0;
    bytes_per_step = (unsigned)(qps * nslices) * 4096u;
  }
  __device__ __forceinline__ v4f record(int j) const { return bld4(rw, vlane, cur + j * 1024); }
  __device__ __forceinline__ void next_quad() {
    cur += 4096;
    if (cur == bytes_per_step) cur = 0;
  }
};

// LDS ring, 2 slots of `chq` quads.  The host picks chq and a per-slice padding so
// that (quads_per_slice + pad) % chq == 0: chunk boundaries then coincide with slice
// ends, i.e. the last commit of a slice sits right before the cell update (whose
// stores would otherwise sit in front of the next commit's in-order vmcnt wait).
// Record r of a chunk is staged by wave r % NW (its (r / NW)-th register, < M).
template <int NW, int M>
struct WStream<NW, M, 1> {
  rsrc_t rw;
  unsigned vlane;
  char* lds_rd;         // smem + lane*16
  char* lds_wr;         // smem + w*1024 + lane*16
  int w;
  int chq;              // quads per chunk
  int ch;               // records per chunk
  int qps4;             // real records per slice
  int vslice4;          // virtual records per slice (incl. padding)
  int recs_per_step;    // real records per step
  int src_slice_base;   // real record index of the slice the next staged chunk belongs to
  int src_off;          // virtual record offset of that chunk inside its slice
  int rq;               // quad index inside the ring, 0 .. 2*chq-1
  int left;             // quads left in the current chunk
  int wslot;
  v4f stg[M];

  __device__ __forceinline__ void issue_loads() {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int r = w + m * NW;              // record inside the chunk
      const int vo = src_off + r;            // virtual offset inside the slice
      if (r < ch && vo < qps4) stg[m] = bld4(rw, vlane, (unsigned)(src_slice_base + vo) * 1024u);
    }
    src_off += ch;
    if (src_off == vslice4) {
      src_off = 0;
      src_slice_base += qps4;
      if (src_slice_base == recs_per_step) src_slice_base = 0;
    }
  }
  __device__ __forceinline__ void commit_and_barrier() {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (w + m * NW < ch) *reinterpret_cast<v4f*>(lds_wr + wslot * (ch * 1024) + m * (NW * 1024)) = stg[m];
    wslot ^= 1;
    // my ring writes have landed and my reads of the previous chunk have returned
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __device__ __forceinline__ void init(const float* wp, int lane, int w_, int qps, int nslices, int chq_, int pad,
                                       char* smem) {
    rw = make_rsrc(wp);
    vlane = lane * 16;
    w = w_;
    lds_rd = smem + lane * 16;
    lds_wr = smem + w_ * 1024 + lane * 16;
    chq = chq_;
    ch = chq_ * 4;
    qps4 = qps * 4;
    vslice4 = (qps + pad) * 4;
    recs_per_step = nslices * qps * 4;
    src_slice_base = 0;
    src_off = 0;
    rq = 0;
    left = chq_;
    wslot = 0;
    issue_loads();          // chunk 0
    commit_and_barrier();   // chunk 0 visible
    issue_loads();          // chunk 1 in flight
  }
  __device__ __forceinline__ v4f record(int j) const {
    return *reinterpret_cast<const v4f*>(lds_rd + rq * 4096 + j * 1024);
  }
  __device__ __forceinline__ void next_quad() {
    rq = (rq + 1 == 2 * chq) ? 0 : rq + 1;
    if (--left == 0) {
      left = chq;
      commit_and_barrier();   // publish the next chunk (loaded one period ago)
      issue_loads();          // and start fetching the one after it
    }
  }
};

template <int H, int NW, int M, int WMODE, bool HAS1, bool HAS2>
__global__ void __launch_bounds__(NW * 64) lstm_rec_kernel(const LstmParams p) {
  constexpr int NS = H / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int task = wg * NW + w;
  int q = task * 16 + n;
  const bool valid = q < p.nseq;
  if (!valid) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  unsigned vo0 = 0, vo1 = 0, vo2 = 0, voo = 0;
  const rsrc_t rx0 = split_addr(p.src0.p, qo * p.src0.so + qi * p.src0.si, 4 * g, vo0);
  const rsrc_t rx1 = HAS1 ? split_addr(p.src1.p, qo * p.src1.so + qi * p.src1.si, 4 * g, vo1) : rx0;
  const rsrc_t rx2 = HAS2 ? split_addr(p.src2.p, qo * p.src2.so + qi * p.src2.si, 0, vo2) : rx0;
  const rsrc_t ro = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo);
  const rsrc_t rc = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                              ((size_t)blockIdx.x * NW + w) * (NS * 1024));
  const unsigned st0 = (unsigned)(p.src0.st * 4), st1 = (unsigned)(p.src1.st * 4);
  const unsigned st2 = HAS2 ? (unsigned)(p.src2.st * 4) : 0u, sto = (unsigned)(p.out_st * 4);
  const unsigned vlane = lane * 16;

  const int nv0 = p.c0 >> 4, ns0 = (p.c0 & 15) >> 2;
  const int nv2 = HAS2 ? p.c2 >> 4 : 0, ns2 = HAS2 ? (p.c2 & 15) >> 2 : 0;
  const bool rev = dir == 1;

  WStream<NW, M, WMODE> ws;
  ws.init(p.wpack[dir], lane, w, p.quads_per_slice, NS, p.chq, p.pad, smem);

  v4f hold[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) hold[s] = v4f{0.f, 0.f, 0.f, 0.f};
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};

  // x-operand prefetch registers.  They always hold the RAW loads of the next two
  // 16-channel blocks this wave will consume (summed only at consumption, so the
  // loads stay in flight behind >= 2 quads of MFMAs); the pipeline runs across
  // slice and step boundaries.
  v4f pa0 = zero4, pb0 = zero4, pa1 = zero4, pb1 = zero4;   // raw blocks 0, 1 of the next slice
  v4f pc0 = zero4, pd0 = zero4, pc1 = zero4, pd1 = zero4;   // raw blocks 2, 3 of the next slice
  {
    const unsigned tt0 = rev ? p.nsteps - 1 : 0;
    if (nv0 > 0) {
      pa0 = bld4(rx0, vo0, tt0 * st0);
      if (HAS1) pb0 = bld4(rx1, vo1, tt0 * st1);
    }
    if (nv0 > 1) {
      pa1 = bld4(rx0, vo0, tt0 * st0 + 64);
      if (HAS1) pb1 = bld4(rx1, vo1, tt0 * st1 + 64);
    }
    if (nv0 > 2) {
      pc0 = bld4(rx0, vo0, tt0 * st0 + 128);
      if (HAS1) pd0 = bld4(rx1, vo1, tt0 * st1 + 128);
    }
    if (nv0 > 3) {
      pc1 = bld4(rx0, vo0, tt0 * st0 + 192);
      if (HAS1) pd1 = bld4(rx1, vo1, tt0 * st1 + 192);
    }
  }

  for (int step = 0; step < p.nsteps; ++step) {
    const unsigned tt = rev ? p.nsteps - 1 - step : step;
    const unsigned ttn = step + 1 < p.nsteps ? (rev ? tt - 1 : tt + 1) : tt;   // prefetch target
    const unsigned o0 = tt * st0, o1 = tt * st1, o2 = tt * st2, oo = tt * sto;

    // the first 4-channel remainder block of each segment is the same for every
    // slice of the step: keep it in a register  (vo0 carries +4g floats: undo 3g)
    float xs0 = 0.f, xs2 = 0.f;
    if (ns0 > 0) {
      xs0 = bld1(rx0, vo0 - 12 * g, o0 + 64 * nv0);
      if (HAS1) xs0 += bld1(rx1, vo1 - 12 * g, o1 + 64 * nv0);
    }
    if (HAS2 && ns2 > 0) xs2 = bld1(rx2, vo2 + 4 * g, o2 + 64 * nv2);
    if (step > 0) {
      // h_{t-1}: each lane re-reads exactly the float4s it stored one step ago
      const unsigned op = (rev ? tt + 1 : tt - 1) * sto;
#pragma unroll
      for (int s = 0; s < NS; ++s) hold[s] = bld4(ro, voo, op + 64 * s);
    }

    for (int s = 0; s < NS; ++s) {
      v4f cprev = zero4;
      if (step > 0) cprev = bld4(rc, vlane, s * 1024);
      v4f xc0 = HAS1 ? pa0 + pb0 : pa0;   // blocks 0, 1: issued during the previous slice
      v4f xc1 = HAS1 ? pa1 + pb1 : pa1;
      pa0 = pc0;                           // blocks 2, 3: issued before the previous cell update,
      pb0 = pd0;                           // i.e. ahead of its stores in the in-order vmcnt queue
      pa1 = pc1;
      pb1 = pd1;

      v4f acc[4];
      acc[0] = ws.record(0);
      acc[1] = ws.record(1);
      acc[2] = ws.record(2);
      acc[3] = ws.record(3);
      ws.next_quad();

      // ---- summed input segment, 16 channels per quad -------------------
      for (int v = 0; v < nv0; v += 2) {
        {
          const v4f a0 = ws.record(0), a1 = ws.record(1), a2 = ws.record(2), a3 = ws.record(3);
          MFMA4(acc, a0, xc0.x);
          MFMA4(acc, a1, xc0.y);
          MFMA4(acc, a2, xc0.z);
          MFMA4(acc, a3, xc0.w);
          ws.next_quad();
        }
        if (v + 1 < nv0) {
          const v4f a0 = ws.record(0), a1 = ws.record(1), a2 = ws.record(2), a3 = ws.record(3);
          MFMA4(acc, a0, xc1.x);
          MFMA4(acc, a1, xc1.y);
          MFMA4(acc, a2, xc1.z);
          MFMA4(acc, a3, xc1.w);
          ws.next_quad();
        }
        if (v + 2 < nv0) {
          xc0 = HAS1 ? pa0 + pb0 : pa0;
          xc1 = HAS1 ? pa1 + pb1 : pa1;
          if (v + 4 < nv0) {
            pa0 = bld4(rx0, vo0, o0 + 64 * (v + 4));
            if (HAS1) pb0 = bld4(rx1, vo1, o1 + 64 * (v + 4));
          }
          if (v + 5 < nv0) {
            pa1 = bld4(rx0, vo0, o0 + 64 * (v + 5));
            if (HAS1) pb1 = bld4(rx1, vo1, o1 + 64 * (v + 5));
          }
        }
      }
      {
        // blocks 0, 1 of the next slice (same x_t) or of the next step
        const unsigned n0 = (s + 1 < NS ? tt : ttn) * st0;
        const unsigned n1 = (s + 1 < NS ? tt : ttn) * st1;
        if (nv0 > 0) {
          pa0 = bld4(rx0, vo0, n0);
          if (HAS1) pb0 = bld4(rx1, vo1, n1);
        }
        if (nv0 > 1) {
          pa1 = bld4(rx0, vo0, n0 + 64);
          if (HAS1) pb1 = bld4(rx1, vo1, n1 + 64);
        }
      }
      for (int u = 0; u < ns0; ++u) {
        float xs = xs0;
        if (u > 0) {   // rare: more than one remainder block, fetched in place
          xs = bld1(rx0, vo0 - 12 * g, o0 + 64 * nv0 + 16 * u);
          if (HAS1) xs += bld1(rx1, vo1 - 12 * g, o1 + 64 * nv0 + 16 * u);
        }
        const v4f a0 = ws.record(0);
        MFMA4(acc, a0, xs);
        ws.next_quad();
      }
      // ---- concatenated input segment -------------------------------------
      for (int v = 0; v < nv2; ++v) {
        const v4f xv = bld4(rx2, vo2 + 16 * g, o2 + 64 * v);
        const v4f a0 = ws.record(0), a1 = ws.record(1), a2 = ws.record(2), a3 = ws.record(3);
        MFMA4(acc, a0, xv.x);
        MFMA4(acc, a1, xv.y);
        MFMA4(acc, a2, xv.z);
        MFMA4(acc, a3, xv.w);
        ws.next_quad();
      }
      for (int u = 0; u < ns2; ++u) {
        float xs = xs2;
        if (u > 0) xs = bld1(rx2, vo2 + 4 * g, o2 + 64 * nv2 + 16 * u);
        const v4f a0 = ws.record(0);
        MFMA4(acc, a0, xs);
        ws.next_quad();
      }
      // ---- recurrent part: B operands are last step's D registers ---------
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const v4f a0 = ws.record(0), a1 = ws.record(1), a2 = ws.record(2), a3 = ws.record(3);
        MFMA4(acc, a0, hold[sp].x);
        MFMA4(acc, a1, hold[sp].y);
        MFMA4(acc, a2, hold[sp].z);
        MFMA4(acc, a3, hold[sp].w);
        ws.next_quad();
      }
      for (int u = 0; u < p.pad; ++u) ws.next_quad();   // ring padding: chunk ends == slice ends
      {
        const unsigned n0 = (s + 1 < NS ? tt : ttn) * st0;
        const unsigned n1 = (s + 1 < NS ? tt : ttn) * st1;
        if (nv0 > 2) {
          pc0 = bld4(rx0, vo0, n0 + 128);
          if (HAS1) pd0 = bld4(rx1, vo1, n1 + 128);
        }
        if (nv0 > 3) {
          pc1 = bld4(rx0, vo0, n0 + 192);
          if (HAS1) pd1 = bld4(rx1, vo1, n1 + 192);
        }
      }
      // ---- cell update (PyTorch gate order i, f, g, o) ----------------------
      const v4f ig = sigmoid4(acc[0]);
      const v4f fg = sigmoid4(acc[1]);
      const v4f gg = tanh4(acc[2]);
      const v4f og = sigmoid4(acc[3]);
      const v4f cn = fg * cprev + ig * gg;
      const v4f hn = og * tanh4(cn);
      bst4(cn, rc, vlane, s * 1024);
      if (valid) bst4(hn, ro, voo, oo + 64 * s);
    }
  }
}

// ---- launcher ----------------------------------------------------------------
template <int H, int NW, int M, int WMODE, bool HAS1, bool HAS2>
int launch_k(const LstmParams& p, int nwg, hipStream_t st) {
  const size_t lds = WMODE ? (size_t)2 * p.chq * 4096 : 0;
  auto k = lstm_rec_kernel<H, NW, M, WMODE, HAS1, HAS2>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_rec_kernel");
  return FNSSL_OK;
}

template <int H, int NW, int M, int WMODE>
int launch_t(const LstmParams& p, bool has1, int nwg, hipStream_t st) {
  const bool has2 = p.c2 > 0;
  if (has1) return has2 ? launch_k<H, NW, M, WMODE, true, true>(p, nwg, st) : launch_k<H, NW, M, WMODE, true, false>(p, nwg, st);
  return has2 ? launch_k<H, NW, M, WMODE, false, true>(p, nwg, st) : launch_k<H, NW, M, WMODE, false, false>(p, nwg, st);
}

// Launch geometries.  NW = waves per workgroup (all share one weight stream),
// M = ring records staged per wave per chunk (chunk <= NW*M records), ring 0 = the
// weight stream is read straight from L1/L2.
struct Variant {
  int NW, M, ring;
};
constexpr int kNumVariants = 8;
const Variant kVariants[kNumVariants + 1] = {
    {0, 0, 0},
    {4, 1, 0},    // 1
    {4, 4, 1},    // 2
    {8, 4, 1},    // 3
    {12, 4, 1},   // 4
    {16, 2, 1},   // 5
    {8, 1, 0},    // 6
    {12, 2, 1},   // 7
    {16, 4, 1},   // 8
};

int env_int(const char* name, int lo, int hi) {
  if (const char* e = getenv(name)) {
    const int v = atoi(e);
    if (v >= lo && v <= hi) return v;
  }
  return 0;
}

int default_variant(int H) {
  // tuning override for experiments: FNSSL_LSTM_VARIANT_H256=3 etc.
  char name[40];
  snprintf(name, sizeof(name), "FNSSL_LSTM_VARIANT_H%d", H);
  if (const int v = env_int(name, 1, kNumVariants)) return v;
  if (H == 256) return 4;
  if (H == 128) return 5;
  return 2;
}

// Ring chunk: the largest chq <= NW*M/4 with (qps + pad) % chq == 0 for a padding
// pad <= 3 quads (chunk boundaries then coincide with slice ends).
void choose_chunk(int qps, const Variant& v, int& chq, int& pad) {
  const int cap = v.NW * v.M / 4;
  int best_c = 1, best_p = 0;
  for (int p = 0; p <= 3; ++p)
    for (int c = cap; c >= 1; --c)
      if ((qps + p) % c == 0) {
        // prefer fewer commits per slice; break ties towards less padding
        if (c > best_c) {
          best_c = c;
          best_p = p;
        }
        break;
      }
  chq = best_c;
  pad = best_p;
  if (const int f = env_int("FNSSL_LSTM_CHQ", 1, cap)) {   // experiments: force a chunk size
    chq = f;
    pad = (f - qps % f) % f;
  }
}

template <int H>
int launch_h(int variant, const LstmParams& p, bool has1, int nwg, hipStream_t st) {
  switch (variant) {
    case 1: return launch_t<H, 4, 1, 0>(p, has1, nwg, st);
    case 2: return launch_t<H, 4, 4, 1>(p, has1, nwg, st);
    case 3: return launch_t<H, 8, 4, 1>(p, has1, nwg, st);
    case 4: return launch_t<H, 12, 4, 1>(p, has1, nwg, st);
    case 5: return launch_t<H, 16, 2, 1>(p, has1, nwg, st);
    case 6: return launch_t<H, 8, 1, 0>(p, has1, nwg, st);
    case 7: return launch_t<H, 12, 2, 1>(p, has1, nwg, st);
    case 8: return launch_t<H, 16, 4, 1>(p, has1, nwg, st);
  }
  fnssl::set_error("lstm: unknown variant %d", variant);
  return FNSSL_E_INVALID;
}

}  // namespace

extern "C" {

size_t fnssl_lstm_packed_floats(int c0, int c2, int hidden) {
  if (hidden <= 0 || hidden % 16 || c0 < 0 || c2 < 0 || (c0 & 3) || (c2 & 3)) return 0;
  return (size_t)(hidden / 16) * quads_per_slice(c0, c2, hidden) * 4 * 256;
}

int fnssl_lstm_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                    int c0, int c2, int H, float* packed) {
  FNSSL_REQUIRE(w_ih && w_hh && b_ih && b_hh && packed, "lstm_pack: null pointer");
  FNSSL_REQUIRE(H > 0 && H % 16 == 0, "lstm_pack: hidden %d must be a positive multiple of 16", H);
  FNSSL_REQUIRE(c0 >= 0 && c2 >= 0 && c0 % 4 == 0 && c2 % 4 == 0 && c0 + c2 > 0,
                "lstm_pack: segment widths (%d, %d) must be multiples of 4", c0, c2);
  const int I = c0 + c2, NS = H / 16;
  const size_t total = fnssl_lstm_packed_floats(c0, c2, H);
  std::memset(packed, 0, total * sizeof(float));
  float* rec = packed;   // 256 floats per record: [lane][gate]
  auto wih = [&](int qg, int unit, int k) { return w_ih[(size_t)(qg * H + unit) * I + k]; };
  auto whh = [&](int qg, int unit, int k) { return w_hh[(size_t)(qg * H + unit) * H + k]; };
  for (int s = 0; s < NS; ++s) {
    // bias quad: record = gate, lane (n, g) component r <-> unit 16s + 4g + r
    for (int qg = 0; qg < 4; ++qg, rec += 256)
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int unit = 16 * s + 4 * (l >> 4) + r;
          rec[l * 4 + r] = b_ih[qg * H + unit] + b_hh[qg * H + unit];
        }
    auto pack_segment = [&](int cbase, int c) {
      const int nv = c >> 4, nsc = (c & 15) >> 2;
      for (int v = 0; v < nv; ++v)
        for (int j = 0; j < 4; ++j, rec += 256)
          for (int l = 0; l < 64; ++l)
            for (int qg = 0; qg < 4; ++qg)
              rec[l * 4 + qg] = wih(qg, 16 * s + (l & 15), cbase + 16 * v + 4 * (l >> 4) + j);
      for (int u = 0; u < nsc; ++u) {
        for (int l = 0; l < 64; ++l)
          for (int qg = 0; qg < 4; ++qg)
            rec[l * 4 + qg] = wih(qg, 16 * s + (l & 15), cbase + 16 * nv + 4 * u + (l >> 4));
        rec += 4 * 256;   // records 1..3 of a scalar quad are padding
      }
    };
    pack_segment(0, c0);
    pack_segment(c0, c2);
    for (int sp = 0; sp < NS; ++sp)
      for (int j = 0; j < 4; ++j, rec += 256)
        for (int l = 0; l < 64; ++l)
          for (int qg = 0; qg < 4; ++qg)
            rec[l * 4 + qg] = whh(qg, 16 * s + (l & 15), 16 * sp + 4 * (l >> 4) + j);
  }
  if ((size_t)(rec - packed) != total) {
    fnssl::set_error("lstm_pack: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

size_t fnssl_lstm_workspace_bytes(int nseq, int hidden, int ndir) {
  if (nseq <= 0 || hidden <= 0 || ndir <= 0) return 0;
  // cell state, one float4 per (lane, slice) per wave; the tail workgroup is
  // padded to a whole workgroup (<= 16 waves), so every variant fits.
  const size_t tasks = (size_t)(nseq + 15) / 16 + 16;
  return tasks * ndir * (size_t)(hidden / 16) * 64 * 16 + 256;
}

int fnssl_lstm_forward(const fnssl_lstm_desc* d, void* stream) {
  FNSSL_REQUIRE(d, "lstm_forward: null descriptor");
  const int H = d->hidden;
  FNSSL_REQUIRE(H == 16 || H == 32 || H == 64 || H == 128 || H == 256,
                "lstm_forward: hidden size %d unsupported (16/32/64/128/256)", H);
  FNSSL_REQUIRE(d->ndir == 1 || d->ndir == 2, "lstm_forward: ndir must be 1 or 2");
  FNSSL_REQUIRE(d->nseq > 0 && d->nsteps > 0 && d->q_inner > 0, "lstm_forward: empty problem");
  FNSSL_REQUIRE(d->c0 >= 0 && d->c2 >= 0 && d->c0 % 4 == 0 && d->c2 % 4 == 0 && d->c0 + d->c2 > 0,
                "lstm_forward: input widths (%d, %d) must be multiples of 4", d->c0, d->c2);
  FNSSL_REQUIRE(d->c0 == 0 || d->src0.p, "lstm_forward: src0 missing");
  FNSSL_REQUIRE(d->c2 == 0 || d->src2.p, "lstm_forward: src2 missing");
  FNSSL_REQUIRE(d->out && d->wpack[0] && (d->ndir == 1 || d->wpack[1]), "lstm_forward: null out/weights");
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  FNSSL_REQUIRE(aligned(d->src0.p) && aligned(d->src1.p) && aligned(d->src2.p) && aligned(d->out) &&
                    aligned(d->wpack[0]) && aligned(d->wpack[1]) && aligned(d->workspace),
                "lstm_forward: pointers must be 16-byte aligned");
  auto mult4 = [](long long v) { return (v & 3) == 0; };
  FNSSL_REQUIRE(mult4(d->src0.so) && mult4(d->src0.si) && mult4(d->src0.st) && mult4(d->out_so) &&
                    mult4(d->out_si) && mult4(d->out_st) &&
                    (!d->src1.p || (mult4(d->src1.so) && mult4(d->src1.si) && mult4(d->src1.st))) &&
                    (!d->src2.p || (mult4(d->src2.so) && mult4(d->src2.si) && mult4(d->src2.st))),
                "lstm_forward: strides must be multiples of 4 floats");
  // buffer addressing: per-wave lane spread + step walk + one row must fit 32 bits
  auto extent_ok = [&](long long so, long long si, long long st, long long width) {
    auto ab = [](long long v) { return v < 0 ? -v : v; };
    const long double e = ((long double)ab(so) + 16.0L * ab(si) + (long double)d->nsteps * ab(st) + width) * 4.0L;
    return so >= 0 && si >= 0 && st >= 0 && e < 4.0e9L;
  };
  FNSSL_REQUIRE(extent_ok(d->src0.so, d->src0.si, d->src0.st, d->c0) &&
                    (!d->src1.p || extent_ok(d->src1.so, d->src1.si, d->src1.st, d->c0)) &&
                    (!d->src2.p || extent_ok(d->src2.so, d->src2.si, d->src2.st, d->c2)) &&
                    extent_ok(d->out_so, d->out_si, d->out_st, 2 * H),
                "lstm_forward: strides must be non-negative and one sequence group must span < 4 GB");
  const size_t need = fnssl_lstm_workspace_bytes(d->nseq, H, d->ndir);
  if (!d->workspace || d->workspace_bytes < need) {
    fnssl::set_error("lstm_forward: workspace %zu < %zu bytes", d->workspace_bytes, need);
    return FNSSL_E_WORKSPACE;
  }
  const int variant = d->variant ? d->variant : default_variant(H);
  FNSSL_REQUIRE(variant >= 1 && variant <= kNumVariants, "lstm_forward: unknown variant %d", variant);
  const Variant& vr = kVariants[variant];

  LstmParams p;
  p.src0 = View{d->src0.p, d->src0.so, d->src0.si, d->src0.st};
  p.src1 = View{d->src1.p, d->src1.so, d->src1.si, d->src1.st};
  p.src2 = View{d->src2.p, d->src2.so, d->src2.si, d->src2.st};
  p.out = d->out;
  p.out_so = d->out_so;
  p.out_si = d->out_si;
  p.out_st = d->out_st;
  p.wpack[0] = d->wpack[0];
  p.wpack[1] = d->wpack[1];
  p.cscratch = d->workspace;
  p.c0 = d->c0;
  p.c2 = d->c2;
  p.nseq = d->nseq;
  p.q_inner = d->q_inner;
  p.nsteps = d->nsteps;
  p.ndir = d->ndir;
  p.quads_per_slice = quads_per_slice(d->c0, d->c2, H);
  p.chq = 0;
  p.pad = 0;
  if (vr.ring) choose_chunk(p.quads_per_slice, vr, p.chq, p.pad);
  const int tasks = (d->nseq + 15) / 16;
  p.wgs_per_dir = (tasks + vr.NW - 1) / vr.NW;
  const int nwg = p.wgs_per_dir * d->ndir;
  const bool has1 = d->src1.p != nullptr && d->c0 > 0;

  const double flops = 2.0 * 4 * H * (double)(d->c0 + d->c2 + H) * d->nseq * (double)d->nsteps * d->ndir;
  static const char* names[5] = {"lstm_h16", "lstm_h32", "lstm_h64", "lstm_h128", "lstm_h256"};
  const int hi = H == 16 ? 0 : H == 32 ? 1 : H == 64 ? 2 : H == 128 ? 3 : 4;
  fnssl::TimedLaunch tl(names[hi], fnssl::as_stream(stream), flops);
  hipStream_t st = fnssl::as_stream(stream);
  switch (H) {
    case 16: return launch_h<16>(variant, p, has1, nwg, st);
    case 32: return launch_h<32>(variant, p, has1, nwg, st);
    case 64: return launch_h<64>(variant, p, has1, nwg, st);
    case 128: return launch_h<128>(variant, p, has1, nwg, st);
    default: return launch_h<256>(variant, p, has1, nwg, st);
  }
}

}  // extern "C"
