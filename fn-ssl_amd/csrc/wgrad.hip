// Weight gradients of one LSTM layer (training row; reference: autograd through nn.LSTM,
// FN-SSL/Lightning/main.py:149-157):
//     dW_ih = dA^T [x0 | x2],   dW_hh = dA^T h_prev,   db_ih = db_hh = sum_r dA[r]
// with R = sequences x steps rows (2.46 M at BASELINE config 4's shard) — ONE hand-written split-K fp32-MFMA product
// per layer instead of the vendor GEMM calls of rounds 1-2 (torch.bmm -> hipBLASLt / rocBLAS through dlopen):
//   * C[m][n] = sum_r dA[r][m] * B[r][n] with B = [x0 | x2 | h_prev] taken IN PLACE from three tensors: h_prev is the
//     layer output shifted by one step inside its sequence (row r - 1 forward, r + 1 reverse, zero at the sequence
//     boundary) — an index shift and a per-row predicate in the loader, never a materialised tensor; both directions
//     of a bidirectional layer are column ranges of the same dA / h tensors and go through the same launch;
//   * a workgroup (8 waves) owns a 256 x 128 tile of C over one SLAB of rows (split-K); a wave owns 64 x 64 of it =
//     16 accumulator quads of v_mfma_f32_16x16x4_f32.  Both operands are row-major in r, which is exactly the MFMA's
//     "k across lane groups" operand shape: a stage = 16 rows of the A panel (256 columns) and of the B panel (128
//     columns) is copied as it lies (float4 per lane, whole 1 KB / 512 B row pieces) into LDS rows padded by 16
//     floats, so that the four k-rows a wave instruction reads fall into four disjoint bank groups (conflict-free
//     ds_read_b32); global loads of stage i + 1 are in flight while stage i is multiplied (register staging, one
//     barrier per stage);
//   * db rides along: the workgroups of the first column tile add up the A panel they stream anyway;
//   * work items are ordered [slab][m tile][n tile] and mapped so that one XCD (blockIdx % 8) takes consecutive items:
//     the tiles that share an A panel or a B panel run on the same L2;
//   * partial tiles go to a workspace [slab][M][N]; a second kernel adds them up in slab order (deterministic, no
//     float atomics) into the caller's gradient (+=: chunks accumulate).
// Roof: fp32 MFMA (157.3 TFLOP/s); algorithmic work 2 * R * M * N flop per layer, compulsory traffic R * (M + N) * 4 B.
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef FNSSL_WGRAD_BK
#define FNSSL_WGRAD_BK 16
#endif
constexpr int kTM = 256, kTN = 128, kBK = FNSSL_WGRAD_BK;   // rows per stage: 16 (two workgroups per CU) or 32 (one; experiment)
constexpr int kRA = kBK / 8, kRB = kBK / 16;                // float4 per thread and stage: A rows ar + 8 i, B rows br + 16 j
constexpr int kLdA = kTM + 16, kLdB = kTN + 16;          // LDS row strides (floats): stride % 64 == 16
constexpr int kThreads = 512;

struct Seg {            // one column segment of B
  const float* p;       // element (r, c) at p[r * ld + c]
  long long ld;
  int cols;             // multiple of 4
  int shift;            // 0, or +-1: h_prev of the forward / reverse direction (row r pairs with row r - shift... see loader)
};

struct WgradParams {
  const float* da;
  long long lda;
  long long rows;
  int nsteps;
  int M;                // ndir * 4H
  int G4;               // 4H: columns of one direction
  int hidden;
  Seg x0, x2;
  const float* h;       // [rows, ndir * H]
  long long ldh;
  int ncat;             // c0 + c2 + H
  int nt0, nt2, nth;    // 128-column tiles per segment
  int mtiles, ntiles, slabs;
  long long rows_per_slab;   // multiple of kBK
  float* part;          // [slabs][M][ncat]
  float* part_db;       // [slabs][M]
  // a 4-channel input segment (block 1: the network input) does not get column tiles of its own: the workgroups of
  // column tile 0 multiply it on the vector pipe from the A panel they stream anyway (32 FMAs per thread and stage)
  const float* xs;      // [rows, 4] (row stride ldxs) or nullptr
  long long ldxs;
  int xs_col;           // its first column in [x0 | x2 | h]
  int timed;            // ablate build: print per-phase cycles of two workgroups
};

__device__ __forceinline__ v4f ldg4(const float* p) { return *reinterpret_cast<const v4f*>(p); }

// SMALL: the instantiation that also carries the 4-channel side product (32 more registers: it is kept out of the common
// kernel, which must stay at <= 128 registers for two resident workgroups = 4 waves per SIMD; the SMALL kernel — block
// 1's two layers — runs one workgroup per CU).
template <bool SMALL>
__device__ __forceinline__ void wgrad_body(const WgradParams& p) {
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  float (*As)[kBK][kLdA] = reinterpret_cast<float (*)[kBK][kLdA]>(wg_smem);
  float (*Bs)[kBK][kLdB] = reinterpret_cast<float (*)[kBK][kLdB]>(wg_smem + 2 * kBK * kLdA);
  // ---- which tile: consecutive items stay on one XCD (workgroups are dealt to the 8 XCDs round-robin)
  const int nitems = p.slabs * p.mtiles * p.ntiles;
  const int per_xcd = (nitems + 7) / 8;
  const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || item >= nitems) return;
  const int nt = item % p.ntiles;
  const int mt = (item / p.ntiles) % p.mtiles;
  const int slab = item / (p.ntiles * p.mtiles);
  const int m0 = mt * kTM;
  const int dir = m0 / p.G4;

  // B segment of this column tile
  const float* bp;
  long long bld;
  int bcols, bc0, shift = 0, ncol0;          // bc0: first column inside the segment; ncol0: first column in [x0|x2|h]
  if (nt < p.nt0) {
    bp = p.x0.p; bld = p.x0.ld; bcols = p.x0.cols; bc0 = nt * kTN; ncol0 = bc0;
  } else if (nt < p.nt0 + p.nt2) {
    bp = p.x2.p; bld = p.x2.ld; bcols = p.x2.cols; bc0 = (nt - p.nt0) * kTN; ncol0 = p.x0.cols + bc0;
  } else {
    bp = p.h + (long long)dir * p.hidden; bld = p.ldh; bcols = p.hidden; bc0 = (nt - p.nt0 - p.nt2) * kTN;
    ncol0 = p.x0.cols + p.x2.cols + bc0;
    shift = dir == 0 ? -1 : 1;               // forward: h of the previous step; reverse: of the next one
  }

  const int tid = threadIdx.x;
  const long long r_begin = (long long)slab * p.rows_per_slab;
  const long long r_end = r_begin + p.rows_per_slab < p.rows ? r_begin + p.rows_per_slab : p.rows;
  const int nstage = (int)((r_end - r_begin + kBK - 1) / kBK);

  // loader roles: A = two float4 per thread (rows ar, ar + 8), B = one float4 per thread
  const int ar = tid >> 6, ac4 = (tid & 63) * 4;
  const int br = tid >> 5, bc4 = (tid & 31) * 4;
  const bool bcol_ok = bc0 + bc4 < bcols;
  const float* a_src = p.da + m0 + ac4;
  const float* b_src = bp + bc0 + bc4;
  const bool do_db = nt == 0;
  const bool do_small = SMALL && do_db && p.xs != nullptr;
  v4f dbacc = {0.f, 0.f, 0.f, 0.f};
  v4f sacc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  // Register staging, TWO stages deep: stage k travels global -> register set k & 1 -> LDS buffer k & 1.  A set is
  // re-requested right after it has been written to LDS, i.e. two stage-times (~4000 cycles of MFMA issue per SIMD)
  // before it is needed again — one stage ahead left the HBM latency of a loaded chip exposed at the stage end.
  struct Regs {
    v4f a[kRA], b[kRB], x[kRA];
  };
  Regs R0, R1;
#pragma unroll
  for (int i = 0; i < kRA; ++i) R0.x[i] = R1.x[i] = v4f{0.f, 0.f, 0.f, 0.f};
  // step index of this thread's B row inside its sequence, advanced by one stage per load (32-bit arithmetic)
  int tb = (int)((r_begin + br) % p.nsteps);
  const int tb_adv = kBK % p.nsteps;
  const int t_bad = shift < 0 ? 0 : p.nsteps - 1;
  auto load_stage = [&](int st, Regs& R) {
    const long long r0 = r_begin + (long long)st * kBK;
    const long long ra = r0 + ar, rb_ = r0 + br;
    const v4f z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kRA; ++i) R.a[i] = ra + 8 * i < r_end ? ldg4(a_src + (ra + 8 * i) * p.lda) : z;
#pragma unroll
    for (int j = 0; j < kRB; ++j) {
      const int tbj = j == 0 ? tb : (tb + 16 * j) % p.nsteps;   // step index of row rb_ + 16 j inside its sequence
      const bool ok = bcol_ok && rb_ + 16 * j < r_end && !(shift != 0 && tbj == t_bad);
      R.b[j] = ok ? ldg4(b_src + (rb_ + 16 * j + shift) * bld) : z;
    }
    if (do_small) {     // the whole wave reads one row: a broadcast load (rows past the end meet zero A values)
#pragma unroll
      for (int i = 0; i < kRA; ++i) R.x[i] = ldg4(p.xs + (ra + 8 * i < p.rows ? ra + 8 * i : p.rows - 1) * p.ldxs);
    }
    tb += tb_adv;
    if (tb >= p.nsteps) tb -= p.nsteps;
  };
  auto store_stage = [&](int buf, const Regs& R) {
#pragma unroll
    for (int i = 0; i < kRA; ++i) *reinterpret_cast<v4f*>(&As[buf][ar + 8 * i][ac4]) = R.a[i];
#pragma unroll
    for (int j = 0; j < kRB; ++j) *reinterpret_cast<v4f*>(&Bs[buf][br + 16 * j][bc4]) = R.b[j];
    if (do_db) {
#pragma unroll
      for (int i = 0; i < kRA; i += 2) dbacc += R.a[i] + R.a[i + 1];
    }
    if (do_small) {
#pragma unroll
      for (int i = 0; i < kRA; i += 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sacc[c] += R.a[i][c] * R.x[i] + R.a[i + 1][c] * R.x[i + 1];
      }
    }
  };

  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w & 3) * 64, wn = (w >> 2) * 64;
  const int l16 = lane & 15, kq = lane >> 4;
  // a column tile that ends inside its segment (the 4-channel inputs of block 1, H = 128 < 2 tiles ...) only
  // multiplies the 16-column sub-tiles that exist: jn = how many of this wave's four
  const int cols_here = bcols - bc0 < kTN ? bcols - bc0 : kTN;
  const int jn = cols_here <= wn ? 0 : ((cols_here - wn + 15) / 16 < 4 ? (cols_here - wn + 15) / 16 : 4);
  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](int buf) {
    // the operands of sub-step kk + 1 are read from LDS BEFORE the MFMAs of sub-step kk (fences: the scheduler otherwise
    // sinks the reads next to their use and every 16 MFMAs wait out an LDS round trip)
    float a[4], b[4], an[4], bn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = As[buf][kq][wm + 16 * i + l16];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kq][wn + 16 * j + l16];
#pragma unroll
    for (int kk = 0; kk < kBK / 4; ++kk) {
      if (kk + 1 < kBK / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) an[i] = As[buf][4 * (kk + 1) + kq][wm + 16 * i + l16];
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = Bs[buf][4 * (kk + 1) + kq][wn + 16 * j + l16];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (jn == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < jn) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 1 < kBK / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = an[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bn[j];
      }
    }
  };
  // stage k lives in register set k & 1 until it is written to LDS buffer k & 1 at the end of stage k - 1
  if (nstage > 0) {
    load_stage(0, R0);
    store_stage(0, R0);
    if (nstage > 1) load_stage(1, R1);
    if (nstage > 2) load_stage(2, R0);
  }
  __syncthreads();
#ifdef FNSSL_BUILD_ABLATE
  unsigned long long tm = 0, ts = 0, tb2 = 0, t0 = __builtin_amdgcn_s_memtime(), tbeg = t0;
#define LAP(X)                                                    \
  do {                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    X += now_ - t0;                                               \
    t0 = now_;                                                    \
  } while (0)
#else
#define LAP(X)
#endif
  for (int st = 0; st < nstage; st += 2) {
    multiply(0);
    LAP(tm);
    if (st + 1 < nstage) store_stage(1, R1);
    if (st + 3 < nstage) load_stage(st + 3, R1);
    LAP(ts);
    __syncthreads();
    LAP(tb2);
    if (st + 1 < nstage) {
      multiply(1);
      LAP(tm);
      if (st + 2 < nstage) store_stage(0, R0);
      if (st + 4 < nstage) load_stage(st + 4, R0);
      LAP(ts);
      __syncthreads();
      LAP(tb2);
    }
  }
#ifdef FNSSL_BUILD_ABLATE
  if (p.timed && (blockIdx.x == 0 || blockIdx.x == 777) && (tid & 63) == 0 && (w == 0 || w == 5))
    printf("wgrad block %d wave %d: %d stages, cycles per stage: total %llu = multiply %llu + stage store / request %llu + barrier %llu\n",
           (int)blockIdx.x, w, nstage, (__builtin_amdgcn_s_memtime() - tbeg) / nstage, tm / nstage, ts / nstage, tb2 / nstage);
#endif
#undef LAP

  // ---- partial tile -> workspace [slab][M][ncat]; D fragment: lane (l16, kq) holds rows 4 kq + r, column l16
  float* out = p.part + ((long long)slab * p.M + m0) * p.ncat + ncol0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wn + 16 * j + l16;
      if (bc0 + n < bcols) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(long long)(wm + 16 * i + 4 * kq + r) * p.ncat + n] = acc[i][j][r];
      }
    }
  if (do_db) {          // column sums of the A panel: 8 row-threads per column quad -> LDS -> one value per column
    float* red = &As[0][0][0];                 // 8 x 256 floats (the stage buffers are free now)
    *reinterpret_cast<v4f*>(red + ar * kTM + ac4) = dbacc;
    __syncthreads();
    if (tid < kTM) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[k * kTM + tid];
      p.part_db[(long long)slab * p.M + m0 + tid] = s;
    }
    if (do_small) {     // the same reduction over the 8 row-threads for the 4-channel products
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<v4f*>(red + ((ar * kTM) + ac4 + c) * 4) = sacc[c];
      __syncthreads();
      if (tid < kTM) {
        v4f s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) s4 += *reinterpret_cast<const v4f*>(red + (k * kTM + tid) * 4);
        float* o = p.part + ((long long)slab * p.M + m0 + tid) * p.ncat + p.xs_col;
        o[0] = s4[0]; o[1] = s4[1]; o[2] = s4[2]; o[3] = s4[3];
      }
    }
  }
}

constexpr size_t kWgradLds = (size_t)2 * kBK * (kLdA + kLdB) * sizeof(float);
#if FNSSL_WGRAD_BK == 16
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) wgrad_kernel(const WgradParams p) {
  wgrad_body<false>(p);
}
#else
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) wgrad_kernel(const WgradParams p) {
  wgrad_body<false>(p);
}
#endif
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 3))) wgrad_small_kernel(const WgradParams p) {
  wgrad_body<true>(p);
}

// grad (+)= sum over slabs, in slab order.  One thread per (m, n) of [M][ncat] and per m for the bias.
struct ReduceParams {
  const float* part;
  const float* part_db;
  int slabs, M, G4, ncat, nin, hidden;       // nin = c0 + c2
  float* g_wih[2];
  float* g_whh[2];
  float* g_bih[2];
  float* g_bhh[2];
};

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const ReduceParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)p.M * p.ncat;
  if (i < total) {
    const int m = (int)(i / p.ncat), n = (int)(i - (long long)m * p.ncat);
    float s = 0.f;
    for (int k = 0; k < p.slabs; ++k) s += p.part[(long long)k * total + i];
    const int dir = m / p.G4, mm = m - dir * p.G4;
    if (n < p.nin)
      p.g_wih[dir][(long long)mm * p.nin + n] += s;
    else
      p.g_whh[dir][(long long)mm * p.hidden + (n - p.nin)] += s;
  } else if (i < total + p.M) {
    const int m = (int)(i - total);
    float s = 0.f;
    for (int k = 0; k < p.slabs; ++k) s += p.part_db[(long long)k * p.M + m];
    const int dir = m / p.G4, mm = m - dir * p.G4;
    p.g_bih[dir][mm] += s;
    p.g_bhh[dir][mm] += s;
  }
}

int plan_slabs(long long rows, int mtiles, int ntiles, long long* rows_per_slab) {
  // enough items for ~3 rounds of two resident workgroups per CU, a whole number of rounds where possible
  const int slots = 2 * fnssl::device_cus();
  const int per_slab = mtiles * ntiles;
  int slabs = (3 * slots + per_slab - 1) / per_slab;
  const long long max_slabs = (rows + 4 * kBK - 1) / (4 * kBK);       // at least 4 stages per slab
  if (slabs > max_slabs) slabs = (int)(max_slabs > 0 ? max_slabs : 1);
  if (slabs > 512) slabs = 512;
  if (slabs < 1) slabs = 1;
  long long rps = (rows + slabs - 1) / slabs;
  rps = (rps + kBK - 1) / kBK * kBK;
  slabs = (int)((rows + rps - 1) / rps);
  *rows_per_slab = rps;
  return slabs;
}

}  // namespace

extern "C" {

size_t fnssl_lstm_weight_grads_workspace_bytes(long long rows, int hidden, int ndir, int c0, int c2) {
  if (rows <= 0 || hidden <= 0 || ndir < 1 || ndir > 2 || c0 < 0 || c2 < 0) return 0;
  const int M = ndir * 4 * hidden, ncat = c0 + c2 + hidden;
  const int mtiles = M / kTM;
  const int ntiles = (c0 == 4 ? 0 : (c0 + kTN - 1) / kTN) + (c2 == 4 && c0 != 4 ? 0 : (c2 + kTN - 1) / kTN) +
                     (hidden + kTN - 1) / kTN;
  long long rps;
  const int slabs = plan_slabs(rows, mtiles > 0 ? mtiles : 1, ntiles, &rps);
  return ((size_t)slabs * M * ncat + (size_t)slabs * M) * sizeof(float) + 256;
}

int fnssl_lstm_weight_grads(const fnssl_wgrad_desc* d, void* stream) {
  FNSSL_REQUIRE(d && d->da && d->h, "lstm_weight_grads: null pointer");
  const int H = d->hidden, nd = d->ndir;
  FNSSL_REQUIRE(nd == 1 || nd == 2, "lstm_weight_grads: ndir %d", nd);
  FNSSL_REQUIRE(H > 0 && (4 * H) % kTM == 0, "lstm_weight_grads: hidden %d (4H must be a multiple of %d)", H, kTM);
  FNSSL_REQUIRE(d->c0 >= 0 && d->c2 >= 0 && d->c0 % 4 == 0 && d->c2 % 4 == 0 && H % 4 == 0 && d->c0 + d->c2 > 0,
                "lstm_weight_grads: input widths (%d, %d) must be multiples of 4", d->c0, d->c2);
  FNSSL_REQUIRE((d->c0 == 0 || d->x0) && (d->c2 == 0 || d->x2), "lstm_weight_grads: missing input tensor");
  FNSSL_REQUIRE(d->nseq > 0 && d->nsteps > 0, "lstm_weight_grads: empty problem");
  FNSSL_REQUIRE(d->lda % 4 == 0 && d->ldh % 4 == 0 && (d->c0 == 0 || d->ldx0 % 4 == 0) && (d->c2 == 0 || d->ldx2 % 4 == 0),
                "lstm_weight_grads: row strides must be multiples of 4 floats");
  FNSSL_REQUIRE(((size_t)d->da | (size_t)d->h | (size_t)d->x0 | (size_t)d->x2) % 16 == 0,
                "lstm_weight_grads: operands must be 16-byte aligned");
  for (int k = 0; k < nd; ++k)
    FNSSL_REQUIRE(d->g_wih[k] && d->g_whh[k] && d->g_bih[k] && d->g_bhh[k], "lstm_weight_grads: null gradient pointer");
  WgradParams p{};
  p.da = d->da;
  p.lda = d->lda;
  p.rows = d->nseq * (long long)d->nsteps;
  p.nsteps = d->nsteps;
  p.M = nd * 4 * H;
  p.G4 = 4 * H;
  p.hidden = H;
  p.x0 = Seg{d->x0, d->ldx0, d->c0, 0};
  p.x2 = Seg{d->x2, d->ldx2, d->c2, 0};
  p.h = d->h;
  p.ldh = d->ldh;
  p.ncat = d->c0 + d->c2 + H;
  p.nt0 = (d->c0 + kTN - 1) / kTN;
  p.nt2 = (d->c2 + kTN - 1) / kTN;
  p.nth = (H + kTN - 1) / kTN;
  if (d->c0 == 4) {             // 4-channel segment: vector-pipe side product in the tile-0 workgroups, no tile
    p.xs = d->x0; p.ldxs = d->ldx0; p.xs_col = 0; p.nt0 = 0;
  } else if (d->c2 == 4) {
    p.xs = d->x2; p.ldxs = d->ldx2; p.xs_col = d->c0; p.nt2 = 0;
  }
  p.mtiles = p.M / kTM;
  p.ntiles = p.nt0 + p.nt2 + p.nth;
  p.slabs = plan_slabs(p.rows, p.mtiles, p.ntiles, &p.rows_per_slab);
#ifdef FNSSL_BUILD_ABLATE
  p.timed = getenv("FNSSL_WGRAD_TIMED") != nullptr;   // per-phase cycle counters: make ABLATE=1 only
#else
  p.timed = false;
#endif
  const size_t need = fnssl_lstm_weight_grads_workspace_bytes(p.rows, H, nd, d->c0, d->c2);
  if (!d->workspace || d->workspace_bytes < need) {
    fnssl::set_error("lstm_weight_grads: workspace %zu < %zu bytes", d->workspace_bytes, need);
    return FNSSL_E_WORKSPACE;
  }
  p.part = reinterpret_cast<float*>(d->workspace);
  p.part_db = p.part + (size_t)p.slabs * p.M * p.ncat;
  hipStream_t st = fnssl::as_stream(stream);
  const int nitems = p.slabs * p.mtiles * p.ntiles;
  const int per_xcd = (nitems + 7) / 8;
  {
    fnssl::TimedLaunch tl(H >= 256 ? "wgrad_h256" : "wgrad_h128", st, 2.0 * (double)p.rows * p.M * p.ncat);
    if (kWgradLds > 48 * 1024) {
      FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradLds));
      FNSSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradLds));
    }
    if (p.xs)
      hipLaunchKernelGGL(wgrad_small_kernel, dim3(per_xcd * 8), dim3(kThreads), kWgradLds, st, p);
    else
      hipLaunchKernelGGL(wgrad_kernel, dim3(per_xcd * 8), dim3(kThreads), kWgradLds, st, p);
    FNSSL_CHECK_LAUNCH("wgrad_kernel");
  }
  ReduceParams r{};
  r.part = p.part;
  r.part_db = p.part_db;
  r.slabs = p.slabs;
  r.M = p.M;
  r.G4 = p.G4;
  r.ncat = p.ncat;
  r.nin = d->c0 + d->c2;
  r.hidden = H;
  for (int k = 0; k < 2; ++k) {
    r.g_wih[k] = d->g_wih[k < nd ? k : 0];
    r.g_whh[k] = d->g_whh[k < nd ? k : 0];
    r.g_bih[k] = d->g_bih[k < nd ? k : 0];
    r.g_bhh[k] = d->g_bhh[k < nd ? k : 0];
  }
  const long long total = (long long)p.M * p.ncat + p.M;
  {
    fnssl::TimedLaunch tl("wgrad_reduce", st);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r);
    FNSSL_CHECK_LAUNCH("wgrad_reduce_kernel");
  }
  return FNSSL_OK;
}

}  // extern "C"
