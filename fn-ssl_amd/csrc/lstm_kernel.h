// Device code and launch templates of the LSTM recurrence kernel (see lstm.hip for
// the formulation).  Included by lstm.hip (host entry points) and by the per-hidden-
// size translation units lstm_h*.hip, which instantiate launch_h<H> so the ~50
// kernel instantiations per H compile in parallel.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "tuning.h"

// HIP's __fadd_rn/__fmul_rn inline to plain fadd/fmul, which clang would still contract into
// fma; every fused multiply-add in this file is written explicitly (__fmaf_rn), so turn the
// implicit contraction off for the whole translation unit.
#pragma clang fp contract(off)

typedef float v4f __attribute__((ext_vector_type(4)));

namespace fnssl_lstm {

struct View {
  const float* p;
  long long so, si, st;
};

struct LstmParams {
  View src0, src1, src2, skip;
  float* out;
  float* out_sum;   // optional second output: h + skip (same strides as out)
  long long out_so, out_si, out_st;
  const float* wpack[2];
  float* cscratch;
  char* cluster_ws;   // hand-off area of the cluster-resident bf16 kernel (lstm_bf16c.h): status, tags, operand records
  float* reserve;     // training forward (MODE bit kSave): gates + cell state per (dir, group, step, slice)
  int ntasks;         // 16-sequence groups per direction (reserve / cell-state indexing)
  int carry;          // streaming: 1 = h_{-1} is the row before `out` (host passes out - out_st), c_{-1} is in cscratch
  int c0, c2;
  int nseq, q_inner, nsteps, ndir;
  int wgs_per_dir;
  int task0, task1;   // this launch covers 16-sequence groups [task0, task1) of every direction
  int quads_per_slice;
  int chq, pad;   // ring chunk (quads) and per-slice padding (quads); 0 for direct variants
  int ablate;     // timing experiments only (env FNSSL_ABLATE): bit flags, see lstm_rec_kernel
  // Guarded fallback launch (behind a cluster-resident kernel of the same call): when non-null the kernel runs only if
  // *guard != 0, i.e. only if the cluster kernel gave up on a hand-off; block 0 then counts the layer in *fallback_count.
  const unsigned* guard = nullptr;
  unsigned* fallback_count = nullptr;
  int dry = 0;    // host only (fnssl_lstm_plan): the launch templates return without launching
};

// first statement of every kernel that can be a guarded fallback (uniform for the whole grid: nobody reaches a barrier)
#define FNSSL_GUARDED_KERNEL(P)                                                                            \
  do {                                                                                                     \
    if ((P).guard) {                                                                                       \
      if (__hip_atomic_load((P).guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;           \
      if ((P).fallback_count && blockIdx.x == 0 && threadIdx.x == 0 && (P).task0 == 0) atomicAdd((P).fallback_count, 1u); \
    }                                                                                                      \
  } while (0)

// cluster-resident bf16 kernels (lstm_bf16c.h): geometry, shared by the kernels and the workspace sizing
__host__ __device__ constexpr int cluster_members(int H) { return H == 256 ? 8 : 4; }        // CUs per cluster (4 gate-row tiles each)
__host__ __device__ constexpr int cluster_parts(int H) { return H == 256 ? 2 : 3; }          // 32-sequence tiles per wave
__host__ __device__ constexpr int cluster_seqs(int H) { return cluster_parts(H) * 256; }     // sequences per cluster: parts x 8 waves x 32
__host__ __device__ constexpr size_t cluster_parity_bytes(int H) {                           // [part][sequence tile 8][block H/16][1 KiB]
  return (size_t)cluster_parts(H) * 8 * (H / 16) * 1024;
}
constexpr int kClusterTagWords = 256;                       // per cluster: [part][wave 8][member 8] words, padded to 1 KiB

// kernel MODE bits
constexpr int kHas1 = 1;   // input segment 0 is src0 + src1
constexpr int kHas2 = 2;   // concatenated segment src2 present
constexpr int kSum = 4;    // epilogue also writes out_sum = h + skip
constexpr int kSave = 8;   // training forward: the epilogue also stores i, f, g, o and c for the backward pass

// Reserve layout (written by the kSave forward, read by lstm_bwd_kernel): lane-private 1 KiB records
//   reserve[(((dir * ntasks + group) * nsteps + t) * NS + slice) * 5 + {i, f, g, o, c}][lane] (float4)
constexpr int kReserveRecs = 5;

// ---- stream geometry (shared by packer and kernel) -------------------------
// per hidden slice (16 units):  quad 0           : 4 bias records (acc init, gate q)
//                               seg0 vec quads    : c0/16 quads, record j <-> k = 16v + 4g + j
//                               seg0 scalar quads : (c0%16)/4 quads, record 0 <-> k = base + g
//                               seg2 vec / scalar : same for the concatenated input
//                               h quads           : H/16 quads, record j <-> k = 16s' + 4g + j
__host__ __device__ inline int quads_per_slice(int c0, int c2, int H) {
  return 1 + (c0 >> 4) + ((c0 & 15) >> 2) + (c2 >> 4) + ((c2 & 15) >> 2) + (H >> 4);
}

// Gate math with every rounding spelled out (no compiler-chosen fma contraction), so that all
// kernel instantiations produce bit-identical results.
__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(-x)));
}
__device__ __forceinline__ float tanh_f(float x) {
  // 1 - 2/(e^{2x}+1): saturates cleanly at +-1, abs error ~1e-7
  return __fmaf_rn(-2.0f, __builtin_amdgcn_rcpf(__fadd_rn(__expf(2.0f * x), 1.0f)), 1.0f);
}
// The 4-wide forms below compute EXACTLY what sigmoid_f / tanh_f / fmul / fadd / fma compute per element (same roundings,
// same constants: __expf(-x) is v_exp_f32(x * -log2e), and exp(2x) is v_exp_f32(x * 2 log2e) — 2x is exact, so scaling
// the constant instead gives the same product), but on packed fp32 instructions (v_pk_mul / v_pk_add / v_pk_fma: two
// elements per issue).  Why it matters (tools/ubench/issue_model.hip, profiles/r04): beside v_mfma_f32_16x16x4_f32 an fp32
// VALU instruction is NOT hidden by other waves' matrix instructions — with four waves per SIMD each v_fma / v_mul / v_add
// costs ~3.5 cycles of matrix-pipe time, a v_pk_* the same for two elements, v_exp / v_rcp ~6.5 — so the cell update is
// paid in full (~480 cycles per 16 x 16 slice) and its instruction COUNT is what there is to save.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f exp2_2(v2f a) { return v2f{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ v2f rcp_2(v2f a) { return v2f{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ v2f sigmoid2(v2f a) {
  const v2f nl2e = {-0x1.715476p+0f, -0x1.715476p+0f};       // -log2(e) as __expf's multiplier (0xbfb8aa3b)
  const v2f one = {1.0f, 1.0f};
  return rcp_2(exp2_2(a * nl2e) + one);
}
__device__ __forceinline__ v2f tanh2(v2f a) {
  const v2f l2e2 = {0x1.715476p+1f, 0x1.715476p+1f};         // 2 log2(e): exp(2x) = exp2(x * 2 log2e)
  const v2f one = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f};
  return __builtin_elementwise_fma(m2, rcp_2(exp2_2(a * l2e2) + one), one);
}
__device__ __forceinline__ v4f join4(v2f lo, v2f hi) { return v4f{lo.x, lo.y, hi.x, hi.y}; }
// PK = false: the same values from one-element instructions — for a kernel at its register limit (packed operands are
// even-aligned register PAIRS; lstm_static2_kernel's allocation went from 2 to 28 spilled registers with them).
template <bool PK = true>
__device__ __forceinline__ v4f sigmoid4(v4f a) {
  if constexpr (PK) return join4(sigmoid2(v2f{a.x, a.y}), sigmoid2(v2f{a.z, a.w}));
  return v4f{sigmoid_f(a.x), sigmoid_f(a.y), sigmoid_f(a.z), sigmoid_f(a.w)};
}
template <bool PK = true>
__device__ __forceinline__ v4f tanh4(v4f a) {
  if constexpr (PK) return join4(tanh2(v2f{a.x, a.y}), tanh2(v2f{a.z, a.w}));
  return v4f{tanh_f(a.x), tanh_f(a.y), tanh_f(a.z), tanh_f(a.w)};
}
template <bool PK = true>
__device__ __forceinline__ v4f cell4(v4f f, v4f c, v4f i, v4f g) {   // f*c + i*g, i*g rounded first
  if constexpr (PK) {
    const v2f lo = __builtin_elementwise_fma(v2f{f.x, f.y}, v2f{c.x, c.y}, v2f{i.x, i.y} * v2f{g.x, g.y});
    const v2f hi = __builtin_elementwise_fma(v2f{f.z, f.w}, v2f{c.z, c.w}, v2f{i.z, i.w} * v2f{g.z, g.w});
    return join4(lo, hi);
  }
  return v4f{__fmaf_rn(f.x, c.x, __fmul_rn(i.x, g.x)), __fmaf_rn(f.y, c.y, __fmul_rn(i.y, g.y)),
             __fmaf_rn(f.z, c.z, __fmul_rn(i.z, g.z)), __fmaf_rn(f.w, c.w, __fmul_rn(i.w, g.w))};
}
template <bool PK = true>
__device__ __forceinline__ v4f mul_rn4(v4f a, v4f b) {
  if constexpr (PK) return join4(v2f{a.x, a.y} * v2f{b.x, b.y}, v2f{a.z, a.w} * v2f{b.z, b.w});
  return v4f{__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y), __fmul_rn(a.z, b.z), __fmul_rn(a.w, b.w)};
}
// h + skip with the reference's rounding (the rounded h is what gets added): no fma contraction
template <bool PK = true>
__device__ __forceinline__ v4f add_rn4(v4f a, v4f b) {
  if constexpr (PK) return join4(v2f{a.x, a.y} + v2f{b.x, b.y}, v2f{a.z, a.w} + v2f{b.z, b.w});
  return v4f{__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w)};
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
// Load that bypasses the per-CU vector L1 (sc1: agent scope) — for data this wave (or a partner wave) STORED
// earlier and reads back through memory: a write does not refresh a line the L1 already holds.
__device__ __forceinline__ v4f bld4_l2(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));
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

// Weight-stream reader.  record(j) reads record j of the current quad,
// peek_next() the first two records of the following quad (the A operands are
// software-pipelined one half-quad ahead so that a wave running alone on its SIMD
// still issues MFMAs back to back), advance() moves on and reports whether a ring
// commit (barrier) happened — after which the peeked records must be re-read.
// WMODE 0: straight from global.  WMODE 1: 2-slot LDS ring.
template <int NW, int M, int WMODE>
struct WStream;

template <int NW, int M>
struct WStream<NW, M, 0> {
  rsrc_t rw;
  unsigned vlane;
  unsigned cur;   // byte offset of the current quad in the stream
  unsigned bytes_per_step;
  bool nobar;
  __device__ __forceinline__ void init(const float* wp, int lane, int /*w*/, int qps, int nslices, int /*chq*/,
                                       int /*pad*/, char* /*smem*/, int /*split*/ = 1, int /*part*/ = 0) {
    rw = make_rsrc(wp);
    vlane = lane * 16;
    cur = 0;
    bytes_per_step = (unsigned)(qps * nslices) * 4096u;
  }
  __device__ __forceinline__ v4f record(int j) const { return bld4(rw, vlane, cur + j * 1024); }
  __device__ __forceinline__ void peek_next(v4f& n0, v4f& n1) const {
    unsigned nx = cur + 4096;
    if (nx == bytes_per_step) nx = 0;
    n0 = bld4(rw, vlane, nx);
    n1 = bld4(rw, vlane, nx + 1024);
  }
  __device__ __forceinline__ bool advance() {
    cur += 4096;
    if (cur == bytes_per_step) cur = 0;
    return false;
  }
};

// LDS ring, 2 slots of `chq` quads.  The host picks chq and a per-slice padding so
// that (quads_per_slice + pad) % chq == 0: chunk boundaries then coincide with slice
// ends.  Record r of a chunk is staged by wave r % NW (its (r / NW)-th register, < M).
template <int NW, int M>
struct WStream<NW, M, 1> {
  rsrc_t rw;
  unsigned vlane;
  char* lds_rd;         // smem + lane*16 (+ part*4096 in split mode)
  char* lds_wr;         // smem + w*1024 + lane*16
  int w;
  int chq;              // (super-)quads per chunk
  int ch;               // records per chunk
  int qps;              // real quads per slice
  int vq;               // virtual quads per slice (incl. padding)
  int nsl;              // slices per step consumed by ONE wave (H/16, or H/16/split)
  int split;            // waves sharing one 16-sequence group (1, 2, 4): the ring then carries "super-quads"
  int sqb;              // bytes per (super-)quad in the ring = split * 4096
  int src_slice;        // local slice the next staged chunk belongs to
  int src_q;            // its first virtual (super-)quad inside the slice
  int rq;               // (super-)quad index inside the ring, 0 .. 2*chq-1
  int left;             // (super-)quads left in the current chunk
  int wslot;
  bool nobar;
  v4f stg[M];

  // Split mode: wave part p of a group works on hidden slices [p*nsl, (p+1)*nsl); a super-quad holds the
  // same quad position of the `split` slices that are in flight together, so all waves of the workgroup
  // still advance through ONE stream in lockstep.  The stream in memory stays in the standard
  // [slice][quad][record] order; only the staging loads compute their source address differently.
  __device__ __forceinline__ void issue_loads() {
    const int per = 4 * split;               // records per super-quad
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int r = w + m * NW;              // record inside the chunk
      const int sq = src_q + r / per;        // virtual quad position inside the slice
      const int pr = (r / 4) % split;        // which part's quad
      if (r < ch && sq < qps)
        stg[m] = bld4(rw, vlane, (unsigned)(((pr * nsl + src_slice) * qps + sq) * 4 + (r & 3)) * 1024u);
    }
    src_q += chq;
    if (src_q == vq) {
      src_q = 0;
      src_slice = src_slice + 1 == nsl ? 0 : src_slice + 1;
    }
  }
  __device__ __forceinline__ void commit_and_barrier() {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (w + m * NW < ch) *reinterpret_cast<v4f*>(lds_wr + wslot * (ch * 1024) + m * (NW * 1024)) = stg[m];
    wslot ^= 1;
    // my ring writes have landed and my reads of the previous chunk have returned
    if (nobar)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __device__ __forceinline__ void init(const float* wp, int lane, int w_, int qps_, int nslices, int chq_, int pad,
                                       char* smem, int split_ = 1, int part = 0) {
    rw = make_rsrc(wp);
    vlane = lane * 16;
    w = w_;
    split = split_;
    sqb = split_ * 4096;
    lds_rd = smem + lane * 16 + part * 4096;
    lds_wr = smem + w_ * 1024 + lane * 16;
    chq = chq_;
    ch = chq_ * 4 * split_;
    qps = qps_;
    vq = qps_ + pad;
    nsl = nslices;
    src_slice = 0;
    src_q = 0;
    rq = 0;
    left = chq_;
    wslot = 0;
    issue_loads();          // chunk 0
    commit_and_barrier();   // chunk 0 visible
    issue_loads();          // chunk 1 in flight
  }
  __device__ __forceinline__ v4f record(int j) const {
    return *reinterpret_cast<const v4f*>(lds_rd + rq * sqb + j * 1024);
  }
  __device__ __forceinline__ void peek_next(v4f& n0, v4f& n1) const {
    // may run ahead of the publishing barrier at a chunk end: the caller re-reads then
    const int nq = (rq + 1 == 2 * chq) ? 0 : rq + 1;
    n0 = *reinterpret_cast<const v4f*>(lds_rd + nq * sqb);
    n1 = *reinterpret_cast<const v4f*>(lds_rd + nq * sqb + 1024);
  }
  __device__ __forceinline__ bool advance() {
    rq = (rq + 1 == 2 * chq) ? 0 : rq + 1;
    if (--left == 0) {
      left = chq;
      commit_and_barrier();   // publish the next chunk (loaded one period ago)
      issue_loads();          // and start fetching the one after it
      return true;
    }
    return false;
  }
};

// ABL = true builds the timing-ablation twin (FNSSL_ABLATE bits: 1 no x loads, 2 no gate
// transcendentals, 4 no c/h stores, 8 no ring barrier, 16 no c load, 32 no h reload); its
// results are wrong by construction and it is never used unless the env var is set.
// SPLIT > 1 (few sequences): SPLIT waves of the workgroup share one 16-sequence group and divide its hidden
// slices among themselves — wave part p computes slices [p*NS/SPLIT, (p+1)*NS/SPLIT) of every step, all of
// them re-read the complete h_{t-1} from the output tensor after a workgroup barrier at the step end.
template <int H, int NW, int M, int WMODE, int MODE, bool ABL = false, int SPLIT = 1>
__global__ void __launch_bounds__(NW * 64, (NW == 4 ? 3 : 1)) lstm_rec_kernel(const LstmParams p) {
  FNSSL_GUARDED_KERNEL(p);
  // (4-wave workgroups are what small launches use, several per CU: keep them at >= 3 waves per SIMD)
  constexpr int NS = H / 16;
  constexpr int NSL = NS / SPLIT;   // slices per wave
  static_assert(NS % SPLIT == 0 && NW % SPLIT == 0 && (SPLIT == 1 || WMODE == 1), "split geometry");
  constexpr bool HAS1 = (MODE & kHas1) != 0, HAS2 = (MODE & kHas2) != 0, SUM = (MODE & kSum) != 0;
  constexpr bool SAVE = (MODE & kSave) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dir = blockIdx.x / p.wgs_per_dir;
  const int wg = blockIdx.x - dir * p.wgs_per_dir;
  const int part = SPLIT > 1 ? w % SPLIT : 0;
  const int task = p.task0 + wg * (NW / SPLIT) + w / SPLIT;
  int q = task * 16 + n;
  const bool valid = q < p.nseq && task < p.task1;
  if (q >= p.nseq) q = p.nseq - 1;
  const long long qo = q / p.q_inner, qi = q - qo * p.q_inner;

  unsigned vo0 = 0, vo1 = 0, vo2 = 0, voo = 0, vok = 0;
  const rsrc_t rx0 = split_addr(p.src0.p, qo * p.src0.so + qi * p.src0.si, 4 * g, vo0);
  const rsrc_t rx1 = HAS1 ? split_addr(p.src1.p, qo * p.src1.so + qi * p.src1.si, 4 * g, vo1) : rx0;
  const rsrc_t rx2 = HAS2 ? split_addr(p.src2.p, qo * p.src2.so + qi * p.src2.si, 0, vo2) : rx0;
  const rsrc_t rsk = SUM ? split_addr(p.skip.p, qo * p.skip.so + qi * p.skip.si, dir * H + 4 * g, vok) : rx0;
  const rsrc_t ro = split_addr(p.out, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo);
  unsigned voo2 = 0;
  const rsrc_t ro2 = SUM ? split_addr(p.out_sum, qo * p.out_so + qi * p.out_si, dir * H + 4 * g, voo2) : ro;
  // cell state: one record per (direction, 16-sequence group, slice) — independent of the launch geometry,
  // so a streaming caller finds it again in the next call; idle tail waves get private dummy slots
  const rsrc_t rc = make_rsrc(reinterpret_cast<const char*>(p.cscratch) +
                              ((size_t)dir * (p.ntasks + 16) + (task < p.task1 ? task : p.ntasks + w)) * (NS * 1024));
  const unsigned cy = (unsigned)p.carry;
  const bool tsave = SAVE && task < p.task1;
  const rsrc_t rres = SAVE ? make_rsrc(reinterpret_cast<const char*>(p.reserve) +
                                       ((size_t)dir * p.ntasks + (tsave ? task : 0)) * p.nsteps *
                                           (size_t)(NS * kReserveRecs * 1024))
                           : rc;
  const unsigned st0 = (unsigned)(p.src0.st * 4), st1 = HAS1 ? (unsigned)(p.src1.st * 4) : 0u;
  const unsigned st2 = HAS2 ? (unsigned)(p.src2.st * 4) : 0u, sto = (unsigned)(p.out_st * 4);
  const unsigned stk = SUM ? (unsigned)(p.skip.st * 4) : 0u;
  const unsigned vlane = lane * 16;

  const int nv0 = p.c0 >> 4, ns0 = (p.c0 & 15) >> 2;
  const int nv2 = HAS2 ? p.c2 >> 4 : 0, ns2 = HAS2 ? (p.c2 & 15) >> 2 : 0;
  const bool rev = dir == 1;

  const int abl = ABL ? p.ablate : 0;
  WStream<NW, M, WMODE> ws;
  ws.nobar = ABL && (abl & 8);
  ws.init(p.wpack[dir], lane, w, p.quads_per_slice, NSL, p.chq, p.pad, smem, SPLIT, part);
  v4f a0 = ws.record(0), a1 = ws.record(1);   // A operands of the current quad's first half

  v4f hold[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) hold[s] = v4f{0.f, 0.f, 0.f, 0.f};
  const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
  v4f acc[4];

// one full quad: 4 records x 4 gates = 16 MFMAs; the next quad's first two records
// are requested between the two halves
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
// a remainder quad: only record 0 is real
#define QUAD1(B0)               \
  do {                          \
    MFMA4(acc, a0, B0);         \
    ws.peek_next(a0, a1);       \
    if (ws.advance()) {         \
      a0 = ws.record(0);        \
      a1 = ws.record(1);        \
    }                           \
  } while (0)

  // x-operand prefetch registers: RAW loads of the next blocks this wave will consume
  // (summed only at consumption, so the loads stay in flight behind >= 2 quads of
  // MFMAs); the pipeline runs across slice and step boundaries.
  v4f pa0 = zero4, pb0 = zero4, pa1 = zero4, pb1 = zero4;   // blocks 0, 1 of the next slice
  v4f pc0 = zero4, pd0 = zero4, pc1 = zero4, pd1 = zero4;   // blocks 2, 3 of the next slice
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
    const unsigned o0 = tt * st0, o1 = tt * st1, o2 = tt * st2, oo = (tt + cy) * sto, ok = tt * stk;

    // the first 4-channel remainder block of each segment is the same for every
    // slice of the step: keep it in a register  (vo0 carries +4g floats: undo 3g)
    float xs0 = 0.f, xs2 = 0.f;
    if (ns0 > 0) {
      xs0 = bld1(rx0, vo0 - 12 * g, o0 + 64 * nv0);
      if (HAS1) xs0 += bld1(rx1, vo1 - 12 * g, o1 + 64 * nv0);
    }
    if (HAS2 && ns2 > 0) xs2 = bld1(rx2, vo2 + 4 * g, o2 + 64 * nv2);
    if ((step > 0 || cy) && !(abl & 32)) {
      // h_{t-1}: each lane re-reads exactly the float4s it stored one step ago (streaming: one call ago)
      const unsigned op = (rev ? tt + 1 : tt - 1 + cy) * sto;
#pragma unroll
      for (int s = 0; s < NS; ++s) hold[s] = bld4(ro, voo, op + 64 * s);
    }

    for (int sl = 0; sl < NSL; ++sl) {
      const int s = SPLIT > 1 ? part * NSL + sl : sl;   // hidden slice this wave computes now
      v4f cprev = zero4, skipv = zero4;
      if ((step > 0 || cy) && !(abl & 16)) cprev = bld4(rc, vlane, s * 1024);
      if (SUM) skipv = bld4(rsk, vok, ok + 64 * s);
      v4f xc0 = HAS1 ? add_rn4(pa0, pb0) : pa0;   // blocks 0, 1: issued during the previous slice
      v4f xc1 = HAS1 ? add_rn4(pa1, pb1) : pa1;
      pa0 = pc0;                           // blocks 2, 3: issued before the previous cell update,
      pa1 = pc1;                           // i.e. ahead of its stores in the in-order vmcnt queue
      if (HAS1) {
        pb0 = pd0;
        pb1 = pd1;
      }

      // ---- bias quad: accumulator init ------------------------------------
      acc[0] = a0;
      acc[1] = a1;
      acc[2] = ws.record(2);
      acc[3] = ws.record(3);
      ws.peek_next(a0, a1);
      if (ws.advance()) {
        a0 = ws.record(0);
        a1 = ws.record(1);
      }

      // ---- summed input segment, 16 channels per quad -------------------
      for (int v = 0; v < nv0; v += 2) {
        QUAD(xc0.x, xc0.y, xc0.z, xc0.w);
        if (v + 1 < nv0) QUAD(xc1.x, xc1.y, xc1.z, xc1.w);
        if (v + 2 < nv0) {
          xc0 = HAS1 ? add_rn4(pa0, pb0) : pa0;
          xc1 = HAS1 ? add_rn4(pa1, pb1) : pa1;
          if (v + 4 < nv0 && !(abl & 1)) {
            pa0 = bld4(rx0, vo0, o0 + 64 * (v + 4));
            if (HAS1) pb0 = bld4(rx1, vo1, o1 + 64 * (v + 4));
          }
          if (v + 5 < nv0 && !(abl & 1)) {
            pa1 = bld4(rx0, vo0, o0 + 64 * (v + 5));
            if (HAS1) pb1 = bld4(rx1, vo1, o1 + 64 * (v + 5));
          }
        }
      }
      const unsigned n0 = (sl + 1 < NSL ? tt : ttn) * st0;
      const unsigned n1 = (sl + 1 < NSL ? tt : ttn) * st1;
      {
        // blocks 0, 1 of the next slice (same x_t) or of the next step
        if (nv0 > 0 && !(abl & 1)) {
          pa0 = bld4(rx0, vo0, n0);
          if (HAS1) pb0 = bld4(rx1, vo1, n1);
        }
        if (nv0 > 1 && !(abl & 1)) {
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
        QUAD1(xs);
      }
      // ---- concatenated input segment -------------------------------------
      for (int v = 0; v < nv2; ++v) {
        const v4f xv = bld4(rx2, vo2 + 16 * g, o2 + 64 * v);
        QUAD(xv.x, xv.y, xv.z, xv.w);
      }
      for (int u = 0; u < ns2; ++u) {
        float xs = xs2;
        if (u > 0) xs = bld1(rx2, vo2 + 4 * g, o2 + 64 * nv2 + 16 * u);
        QUAD1(xs);
      }
      // ---- recurrent part: B operands are last step's D registers ---------
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) QUAD(hold[sp].x, hold[sp].y, hold[sp].z, hold[sp].w);
      for (int u = 0; u < p.pad; ++u) {   // ring padding: chunk ends == slice ends
        ws.peek_next(a0, a1);
        if (ws.advance()) {
          a0 = ws.record(0);
          a1 = ws.record(1);
        }
      }
      // blocks 2, 3 of the next slice: requested before this slice's stores
      if (nv0 > 2 && !(abl & 1)) {
        pc0 = bld4(rx0, vo0, n0 + 128);
        if (HAS1) pd0 = bld4(rx1, vo1, n1 + 128);
      }
      if (nv0 > 3 && !(abl & 1)) {
        pc1 = bld4(rx0, vo0, n0 + 192);
        if (HAS1) pd1 = bld4(rx1, vo1, n1 + 192);
      }
      // ---- cell update (PyTorch gate order i, f, g, o) ----------------------
      v4f cn, hn;
      if (ABL && (abl & 2)) {
        cn = acc[1] + cprev + acc[0];
        hn = acc[3] + acc[2];
      } else {
        const v4f ig = sigmoid4(acc[0]);
        const v4f fg = sigmoid4(acc[1]);
        const v4f gg = tanh4(acc[2]);
        const v4f og = sigmoid4(acc[3]);
        cn = cell4(fg, cprev, ig, gg);
        hn = mul_rn4(og, tanh4(cn));
        // make the rounded h opaque: h + skip below must add the ROUNDED h (what the reference
        // adds), not become fma(o, tanh c, skip)
        asm("" : "+v"(hn.x), "+v"(hn.y), "+v"(hn.z), "+v"(hn.w));
        if (SAVE && tsave) {
          const unsigned rb = (tt * NS + s) * (kReserveRecs * 1024);
          bst4(ig, rres, vlane, rb);
          bst4(fg, rres, vlane, rb + 1024);
          bst4(gg, rres, vlane, rb + 2048);
          bst4(og, rres, vlane, rb + 3072);
          bst4(cn, rres, vlane, rb + 4096);
        }
      }
      if (!(abl & 4)) {
        bst4(cn, rc, vlane, s * 1024);
        if (valid) {
          bst4(hn, ro, voo, oo + 64 * s);
          if (SUM) bst4(add_rn4(hn, skipv), ro2, voo2, oo + 64 * s);
        }
      } else {
        asm volatile("" ::"v"(cn), "v"(hn));
      }
    }
    if (SPLIT > 1) {
      // the partner waves read my h slices at the start of the next step: stores performed, then meet
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
#undef QUAD
#undef QUAD1
}

// ---- launcher ----------------------------------------------------------------
template <int H, int NW, int M, int WMODE, int MODE, bool ABL = false, int SPLIT = 1>
int launch_k(const LstmParams& p, int nwg, hipStream_t st) {
  if (p.dry) return FNSSL_OK;   // fnssl_lstm_plan: report the family, launch nothing
  const size_t lds = WMODE ? (size_t)2 * p.chq * SPLIT * 4096 : 0;
  auto k = lstm_rec_kernel<H, NW, M, WMODE, MODE, ABL, SPLIT>;
  if (lds > 48 * 1024)
    FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, st, p);
  FNSSL_CHECK_LAUNCH("lstm_rec_kernel");
  return FNSSL_OK;
}

template <int H, int NW, int M, int WMODE>
int launch_t(const LstmParams& p, int mode, int nwg, hipStream_t st) {
  switch (mode) {
    case 0: return launch_k<H, NW, M, WMODE, 0>(p, nwg, st);
    case kHas1: return launch_k<H, NW, M, WMODE, kHas1>(p, nwg, st);
    case kHas2: return launch_k<H, NW, M, WMODE, kHas2>(p, nwg, st);
    case kSum:
      if constexpr (H >= 128 && WMODE == 1 && ((NW == 12 && M == 4) || (NW == 16 && M == 2))) {
#ifdef FNSSL_BUILD_ABLATE
        if (p.ablate) return launch_k<H, NW, M, WMODE, kSum, true>(p, nwg, st);
#endif
      }
      return launch_k<H, NW, M, WMODE, kSum>(p, nwg, st);
    case kHas2 | kSum: return launch_k<H, NW, M, WMODE, kHas2 | kSum>(p, nwg, st);
    case kHas1 | kHas2: return launch_k<H, NW, M, WMODE, kHas1 | kHas2>(p, nwg, st);
  }
  fnssl::set_error("lstm: input combination %d not built (src1 together with out_sum)", mode);
  return FNSSL_E_INVALID;
}

// Launch geometries.  NW = waves per workgroup (all share one weight stream),
// M = ring records staged per wave per chunk (chunk <= NW*M records), ring 0 = the
// weight stream is read straight from L1/L2.
struct Variant {
  int NW, M, ring;
};
constexpr int kNumVariants = 11;
constexpr Variant kVariants[kNumVariants + 1] = {
    {0, 0, 0},
    {4, 1, 0},    // 1
    {4, 4, 1},    // 2
    {8, 4, 1},    // 3
    {12, 4, 1},   // 4
    {16, 2, 1},   // 5
    {8, 1, 0},    // 6
    {12, 2, 1},   // 7
    {16, 4, 1},   // 8
    {13, 2, 1},   // 9   (odd sizes: per-round wave counts of the launch planner)
    {14, 2, 1},   // 10
    {15, 2, 1},   // 11
};

#ifdef FNSSL_BUILD_ABLATE
// `make ABLATE=1` builds only (libfnssl_hip_abl.so: timing twins that skip work): their knobs are environment variables.
// The shipping library reads NO environment variable — it does not even contain this function — see tuning.h.
inline int env_int(const char* name, int lo, int hi) {
  if (const char* e = getenv(name)) {
    const int v = atoi(e);
    if (v >= lo && v <= hi) return v;
  }
  return 0;
}
#endif

// tuning override for experiments: fnssl_tuning.knob[FNSSL_TUNE_LSTM_VARIANT_H256] = 3 etc. (0 = none)
inline int default_variant_override(int H) {
  if (H == 128) return fnssl::tune(FNSSL_TUNE_LSTM_VARIANT_H128, 1, kNumVariants);
  if (H == 256) return fnssl::tune(FNSSL_TUNE_LSTM_VARIANT_H256, 1, kNumVariants);
  return 0;
}

inline int default_variant(int H) {
  if (const int v = default_variant_override(H)) return v;
  if (H == 256) return 4;
  if (H == 128) return 5;
  return 2;
}

// Ring chunk: the largest chq <= NW*M/4 with (qps + pad) % chq == 0 for a padding
// pad <= 3 quads (chunk boundaries then coincide with slice ends).
inline void choose_chunk(int qps, const Variant& v, int& chq, int& pad, int split = 1, int max_chq = 0) {
  int cap = v.NW * v.M / (4 * split);   // split mode: the ring carries super-quads of `split` quads
  if (max_chq > 0 && cap > max_chq) cap = max_chq;   // LDS budget of the launch (several workgroups per CU)
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
  if (const int f = fnssl::tune(FNSSL_TUNE_LSTM_CHQ, 1, cap)) {   // experiments: force a chunk size
    chq = f;
    pad = (f - qps % f) % f;
  }
}

// Split geometries for launches with fewer 16-sequence groups than SIMDs (single utterances, streaming
// chunks): 2 waves per group in 4-wave workgroups, or 4 waves per group in 8-wave workgroups.
template <int H, int NW, int M, int SPLIT>
int launch_split_t(const LstmParams& p, int mode, int nwg, hipStream_t st) {
  switch (mode) {
    case 0: return launch_k<H, NW, M, 1, 0, false, SPLIT>(p, nwg, st);
    case kHas1: return launch_k<H, NW, M, 1, kHas1, false, SPLIT>(p, nwg, st);
    case kHas2: return launch_k<H, NW, M, 1, kHas2, false, SPLIT>(p, nwg, st);
    case kSum: return launch_k<H, NW, M, 1, kSum, false, SPLIT>(p, nwg, st);
    case kHas2 | kSum: return launch_k<H, NW, M, 1, kHas2 | kSum, false, SPLIT>(p, nwg, st);
    case kHas1 | kHas2: return launch_k<H, NW, M, 1, kHas1 | kHas2, false, SPLIT>(p, nwg, st);
  }
  fnssl::set_error("lstm: input combination %d not built (src1 together with out_sum)", mode);
  return FNSSL_E_INVALID;
}

template <int H>
int launch_split_h(int split, const LstmParams& p, int mode, int nwg, hipStream_t st) {
  if (split == 2) return launch_split_t<H, 4, 4, 2>(p, mode, nwg, st);
  if (split == 4) return launch_split_t<H, 8, 8, 4>(p, mode, nwg, st);
  fnssl::set_error("lstm: unsupported split %d", split);
  return FNSSL_E_INVALID;
}

template <int H>
int launch_h(int variant, const LstmParams& p, int mode, int nwg, hipStream_t st) {
  switch (variant) {
    case 1: return launch_t<H, 4, 1, 0>(p, mode, nwg, st);
    case 2: return launch_t<H, 4, 4, 1>(p, mode, nwg, st);
    case 3: return launch_t<H, 8, 4, 1>(p, mode, nwg, st);
    case 4: return launch_t<H, 12, 4, 1>(p, mode, nwg, st);
    case 5: return launch_t<H, 16, 2, 1>(p, mode, nwg, st);
    case 6: return launch_t<H, 8, 1, 0>(p, mode, nwg, st);
    case 7: return launch_t<H, 12, 2, 1>(p, mode, nwg, st);
    case 8: return launch_t<H, 16, 4, 1>(p, mode, nwg, st);
    case 9: return launch_t<H, 13, 2, 1>(p, mode, nwg, st);
    case 10: return launch_t<H, 14, 2, 1>(p, mode, nwg, st);
    case 11: return launch_t<H, 15, 2, 1>(p, mode, nwg, st);
  }
  fnssl::set_error("lstm: unknown variant %d", variant);
  return FNSSL_E_INVALID;
}


}  // namespace fnssl_lstm
