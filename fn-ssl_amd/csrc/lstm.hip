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
#include "tuning.h"

#include "lstm_static.h"
#include "lstm_bf16.h"
#include "lstm_bf16w.h"
#include "lstm_split_static.h"

using namespace fnssl_lstm;

namespace fnssl_lstm {
extern template int launch_h<16>(int, const LstmParams&, int, int, hipStream_t);
extern template int launch_h<32>(int, const LstmParams&, int, int, hipStream_t);
extern template int launch_h<64>(int, const LstmParams&, int, int, hipStream_t);
extern template int launch_h<128>(int, const LstmParams&, int, int, hipStream_t);
extern template int launch_h<256>(int, const LstmParams&, int, int, hipStream_t);
extern template int launch_split_h<128>(int, const LstmParams&, int, int, hipStream_t);
extern template int launch_split_h<256>(int, const LstmParams&, int, int, hipStream_t);
int forward_save(LstmParams p, int H, int mode, hipStream_t st);   // lstm_train.hip
int forward_bf16(LstmParams p, int H, hipStream_t st);             // lstm_bf16.hip
bool f32c_handles(const LstmParams& p, int H, int mode);           // lstm_f32c.hip
int forward_f32c(LstmParams p, int H, int mode, hipStream_t st);          // FNSSL_OK, kNoCluster (not co-resident: caller takes the rounds) or an error
}  // namespace fnssl_lstm

// ---- launch planner (host) -------------------------------------------------------------------------------------
// Rounds (one launch each, one workgroup per CU) for `tasks` 16-sequence groups per direction on `ncu` CUs: the CHEAPEST
// sequence of supported workgroup sizes that covers W = ceil(tasks * ndir / ncu) waves per CU, largest first.
static int plan_lstm_rounds(int H, int tasks, int ndir, int ncu, std::vector<int>& nw_out, std::vector<int>& var_out) {
  struct Sup {
    int nw, variant;
  };
  static const Sup sup128[] = {{4, 2}, {8, 3}, {12, 4}, {13, 9}, {14, 10}, {15, 11}, {16, 5}};
  static const Sup sup256[] = {{4, 2}, {8, 3}, {12, 4}};
  const Sup* sup = H == 128 ? sup128 : sup256;
  const int nsup = H == 128 ? 7 : 3;
  const long long total = (long long)tasks * ndir;
  const int W = (int)((total + ncu - 1) / ncu);            // waves per CU if spread evenly
  // + a fixed cost per round (a launch of 256-300 dependent steps) and a slightly convex per-wave term (ties go to
  // the evenest split: 15 + 14 measured 94.2 ms against 95.5 for 16 + 13)
  auto cost = [](int nw) { return 2.0 + 12.0 * ((nw + 3) / 4) + 0.06 * nw + 0.002 * nw * nw; };
  // best[w] = cheapest cost to cover w waves per CU; first[w] = a round of that solution
  constexpr int kMaxW = 4096;
  FNSSL_REQUIRE(W >= 1 && W <= kMaxW, "lstm_forward: %d waves per CU is beyond the launch planner", W);
  std::vector<double> best(W + 1, 0.0);
  std::vector<int> first(W + 1, 0);
  for (int w = 1; w <= W; ++w) {
    best[w] = 1e30;
    for (int i = 0; i < nsup; ++i) {
      const int rest = w - sup[i].nw > 0 ? w - sup[i].nw : 0;
      const double c = cost(sup[i].nw) + best[rest];
      if (c < best[w] - 1e-9 || (c < best[w] + 1e-9 && sup[i].nw > sup[first[w]].nw)) {
        best[w] = c;
        first[w] = i;
      }
    }
  }
  std::vector<int> seq;
  for (int w = W; w > 0;) {
    seq.push_back(first[w]);
    w -= sup[first[w]].nw;
  }
  std::sort(seq.begin(), seq.end(), [&](int a, int b) { return sup[a].nw > sup[b].nw; });   // the partial round last
  nw_out.clear();
  var_out.clear();
  for (int i : seq) {
    nw_out.push_back(sup[i].nw);
    var_out.push_back(sup[i].variant);
  }
  return FNSSL_OK;
}

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

// bytes of cell-state scratch at the front of the workspace (a multiple of 256)
static size_t cell_scratch_bytes(int nseq, int hidden, int ndir) {
  // one float4 per (lane, slice) per wave; the tail workgroup is padded to a whole workgroup (<= 16 waves), so every
  // variant fits
  const size_t tasks = (size_t)(nseq + 15) / 16 + 16;
  return tasks * ndir * (size_t)(hidden / 16) * 64 * 16 + 256;
}
// the pair-interleaved copy of a weight stream (lstm_static3.h) lives behind the cell state: room for the largest
// stream of the hidden size (c0 + c2 <= 272 channels) per direction
static size_t pair_stream_bytes(int hidden, int ndir) {
  return hidden == 256 ? (size_t)ndir * (hidden / 16) * quads_per_slice(256, 16, hidden) * 4096 : 0;
}

// the hand-off area of the cluster-resident bf16 kernel (lstm_bf16c.h) lives behind that: status word, tags, operand records
// (bf16 "wide" calls: tags + two parities of operand records, ~513 B per sequence; fp32 calls: the cluster kernel of
//  lstm_f32c.h hands h_t over through the output tensor and needs the status word + one tag word per member (H / 16) per
//  16-sequence group and direction, rounded up per cluster: 2 - 4 B per sequence)
static size_t cluster_bytes(int nseq, int hidden, int ndir, int precision) {
  if (hidden != 256 && hidden != 128) return 0;
  if (precision == FNSSL_PRECISION_BF16W) {
    // clusters of the call: full ones (8 NP tiles of 32 sequences) — or, H = 128, up to tiles / 17 smaller ones (forward_bf16c)
    const size_t tiles = (size_t)(nseq + 31) / 32;
    const size_t ncl = (size_t)ndir * std::max<size_t>((nseq + cluster_seqs(hidden) - 1) / cluster_seqs(hidden), hidden == 128 ? tiles / 17 : 0);
    return 256 + ncl * (kClusterTagWords * 4) + ncl * 2 * cluster_parity_bytes(hidden);
  }
  if (precision != FNSSL_PRECISION_FP32) return 256;
  const size_t groups = (size_t)(nseq + 15) / 16;
  return 256 + (groups + 512) * ndir * (hidden / 16) * sizeof(unsigned);   // one tag word per (group, member); +512: the last cluster's groups are rounded up
}

// ... and behind that, for the one shape whose STREAMING calls (carry_state) the cluster-resident fp32 kernel takes (H = 256,
// one direction): a snapshot of the carried cell state.  lstm_f32c_kernel updates c in place every step, so a launch that gives
// up has already overwritten c_{-1}; the guarded fallback of the same call restarts from the snapshot (restore_cell_kernel).
static size_t carry_backup_bytes(int nseq, int hidden, int ndir, int precision) {
  return (precision == FNSSL_PRECISION_FP32 && hidden == 256 && ndir == 1) ? cell_scratch_bytes(nseq, hidden, ndir) : 0;
}

int fnssl_lstm_plan_rounds(int hidden, int nseq, int ndir, int ncu, int* waves_per_wg, int cap) {
  FNSSL_REQUIRE((hidden == 128 || hidden == 256) && nseq > 0 && (ndir == 1 || ndir == 2) && ncu > 0 && waves_per_wg && cap > 0,
                "lstm_plan_rounds: bad arguments");
  std::vector<int> nw, var;
  const int rc = plan_lstm_rounds(hidden, (nseq + 15) / 16, ndir, ncu, nw, var);
  if (rc != FNSSL_OK) return rc;
  for (size_t i = 0; i < nw.size() && (int)i < cap; ++i) waves_per_wg[i] = nw[i];
  return (int)nw.size();
}

size_t fnssl_lstm_workspace_bytes_ex(int nseq, int hidden, int ndir, int precision) {
  if (nseq <= 0 || hidden <= 0 || ndir <= 0) return 0;
  return cell_scratch_bytes(nseq, hidden, ndir) + pair_stream_bytes(hidden, ndir) + cluster_bytes(nseq, hidden, ndir, precision) +
         carry_backup_bytes(nseq, hidden, ndir, precision);
}

size_t fnssl_lstm_workspace_bytes(int nseq, int hidden, int ndir) {   // sufficient for every precision
  return fnssl_lstm_workspace_bytes_ex(nseq, hidden, ndir, FNSSL_PRECISION_BF16W);
}

namespace {
// standard stream [slice][quad][4 records] -> pair-interleaved [slice pair][quad][slice in pair][4 records]; one thread
// per float4 of the stream
__global__ void __launch_bounds__(256) pair_stream_kernel(const float4* __restrict__ in, int qps, long long n4,
                                                          float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int l = (int)(i & 63);                      // float4 inside the 1-KiB record
  const long long rec = i >> 6;                      // output record index = ((p * qps + q) * 2 + j) * 4 + r
  const int r = (int)(rec & 3), j = (int)((rec >> 2) & 1);
  const long long pq = rec >> 3;
  const int q = (int)(pq % qps);
  const long long pp = pq / qps;
  const long long src = (((2 * pp + j) * qps + q) * 4 + r) * 64 + l;
  out[i] = in[src];
}

// standard stream [slice][quad][4 records] -> quad-interleaved [slice quad][quad][slice in quad][4 records] (lstm_static4.h)
__global__ void __launch_bounds__(256) quad_stream_kernel(const float4* __restrict__ in, int qps, long long n4,
                                                          float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int l = (int)(i & 63);                      // float4 inside the 1-KiB record
  const long long rec = i >> 6;                      // output record index = ((p * qps + q) * 4 + j) * 4 + r
  const int r = (int)(rec & 3), j = (int)((rec >> 2) & 3);
  const long long pq = rec >> 4;
  const int q = (int)(pq % qps);
  const long long pp = pq / qps;
  const long long src = (((4 * pp + j) * qps + q) * 4 + r) * 64 + l;
  out[i] = in[src];
}

// guarded fallback of a streaming (carry) call, first step: put back the cell state the aborted cluster kernel has advanced.
// Returns at once unless the cluster kernel left a non-zero status word.
__global__ void __launch_bounds__(256) restore_cell_kernel(const unsigned* __restrict__ guard, const float4* __restrict__ backup,
                                                           float4* __restrict__ cell, long long n4) {
  if (__hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) cell[i] = backup[i];
}
}  // namespace

int fnssl_lstm_cluster_status(const void* workspace, size_t workspace_bytes, int nseq, int hidden, int ndir, void* stream,
                              unsigned* status) {
  FNSSL_REQUIRE(workspace && status && nseq > 0 && ndir > 0, "lstm_cluster_status: null pointer / empty problem");
  FNSSL_REQUIRE(workspace_bytes >= fnssl_lstm_workspace_bytes_ex(nseq, hidden, ndir, FNSSL_PRECISION_FP32) && (hidden == 128 || hidden == 256),
                "lstm_cluster_status: not a workspace of a cluster-kernel shape (hidden %d)", hidden);
  const char* word = reinterpret_cast<const char*>(workspace) + cell_scratch_bytes(nseq, hidden, ndir) + pair_stream_bytes(hidden, ndir);
  hipStream_t st = fnssl::as_stream(stream);
  FNSSL_HIP(hipMemcpyAsync(status, word, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  FNSSL_HIP(hipStreamSynchronize(st));
  return FNSSL_OK;
}

}  // extern "C"

// dry = true (fnssl_lstm_plan): every decision of the real call is taken, nothing is enqueued; *family / *rounds report
// the kernel family and its number of launches
static int lstm_forward_impl(const fnssl_lstm_desc* d, void* stream, bool dry, int* family, int* rounds) {
  FNSSL_REQUIRE(d, "lstm_forward: null descriptor");
  fnssl::TuningScope tuning_of_this_call(d->tuning);
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
  FNSSL_REQUIRE(!d->out_sum || d->skip.p, "lstm_forward: out_sum needs a skip view");
  // no input may be an output: the recurrence re-reads h_{t-1} from `out`, and the guarded fallback kernels behind a cluster
  // kernel that gave up recompute the layer from the inputs — which the first attempt must not have overwritten
  if (!dry) {   // (fnssl_lstm_plan launches nothing: a query may describe the shapes with one buffer)
    const void* outs[2] = {d->out, d->out_sum};
    const void* ins[4] = {d->src0.p, d->src1.p, d->src2.p, d->skip.p};
    for (const void* o : outs)
      for (const void* i : ins) FNSSL_REQUIRE(!o || o != i, "lstm_forward: an input tensor aliases an output tensor");
  }
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  FNSSL_REQUIRE(aligned(d->src0.p) && aligned(d->src1.p) && aligned(d->src2.p) && aligned(d->out) &&
                    aligned(d->wpack[0]) && aligned(d->wpack[1]) && aligned(d->workspace) && aligned(d->skip.p) &&
                    aligned(d->out_sum),
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
                    (!d->out_sum || (extent_ok(d->skip.so, d->skip.si, d->skip.st, 2 * H) && mult4(d->skip.so) &&
                                     mult4(d->skip.si) && mult4(d->skip.st))) &&
                    extent_ok(d->out_so, d->out_si, d->out_st, 2 * H),
                "lstm_forward: strides must be non-negative and one sequence group must span < 4 GB");
  const size_t need = fnssl_lstm_workspace_bytes_ex(d->nseq, H, d->ndir, d->precision);
  if (!d->workspace || d->workspace_bytes < need) {
    fnssl::set_error("lstm_forward: workspace %zu < %zu bytes", d->workspace_bytes, need);
    return FNSSL_E_WORKSPACE;
  }
  LstmParams p;
  p.src0 = View{d->src0.p, d->src0.so, d->src0.si, d->src0.st};
  p.src1 = View{d->src1.p, d->src1.so, d->src1.si, d->src1.st};
  p.src2 = View{d->src2.p, d->src2.so, d->src2.si, d->src2.st};
  p.skip = View{d->skip.p, d->skip.so, d->skip.si, d->skip.st};
  FNSSL_REQUIRE(!d->carry_state || (d->ndir == 1 && !d->reserve),
                "lstm_forward: carry_state needs a uni-directional layer (and is not a training mode)");
  p.carry = d->carry_state ? 1 : 0;
  // streaming: the kernel addresses the output one row down so that "step -1" is the caller's h_{-1} row
  p.out = d->carry_state ? d->out - d->out_st : d->out;
  p.out_sum = (d->carry_state && d->out_sum) ? d->out_sum - d->out_st : d->out_sum;
  p.out_so = d->out_so;
  p.out_si = d->out_si;
  p.out_st = d->out_st;
  p.wpack[0] = d->wpack[0];
  p.wpack[1] = d->wpack[1];
  p.cscratch = d->workspace;
  p.cluster_ws = reinterpret_cast<char*>(d->workspace) + cell_scratch_bytes(d->nseq, H, d->ndir) + pair_stream_bytes(H, d->ndir);
  p.reserve = d->reserve;
  p.ntasks = (d->nseq + 15) / 16;
  p.c0 = d->c0;
  p.c2 = d->c2;
  p.nseq = d->nseq;
  p.q_inner = d->q_inner;
  p.nsteps = d->nsteps;
  p.ndir = d->ndir;
  p.quads_per_slice = quads_per_slice(d->c0, d->c2, H);
#ifdef FNSSL_BUILD_ABLATE
  p.ablate = env_int("FNSSL_ABLATE", 1, 255);   // timing experiments: twin kernels that skip work (wrong results)
#else
  p.ablate = 0;                                 // the shipping library contains no ablation twins (make ABLATE=1)
#endif
  p.dry = dry ? 1 : 0;
  p.fallback_count = d->fallback_count;
  int nlaunch = 0;
  bool guarded = false;   // true: what follows is the guarded fallback of a cluster kernel (family already reported)
  auto report = [&](int f) {
    if (family && !guarded) *family = f;
  };
  const int tasks = (d->nseq + 15) / 16;
  const int mode = ((d->src1.p != nullptr && d->c0 > 0) ? kHas1 : 0) | (d->c2 > 0 ? kHas2 : 0) |
                   (d->out_sum ? kSum : 0);

  const double flops = 2.0 * 4 * H * (double)(d->c0 + d->c2 + H) * d->nseq * (double)d->nsteps * d->ndir;
  static const char* names[5] = {"lstm_h16", "lstm_h32", "lstm_h64", "lstm_h128", "lstm_h256"};
  const int hi = H == 16 ? 0 : H == 32 ? 1 : H == 64 ? 2 : H == 128 ? 3 : 4;
  hipStream_t st = fnssl::as_stream(stream);
  fnssl::TimedLaunch tl(dry ? nullptr : names[hi], st, flops);

  if (d->precision == FNSSL_PRECISION_BF16) {   // bf16 MFMA operands (weights packed by fnssl_lstm_pack_bf16)
    FNSSL_REQUIRE(!(mode & (kHas1 | kSum)) && !d->reserve && !d->carry_state && d->c0 % 16 == 0 && d->c2 % 16 == 0,
                  "lstm_forward: the bf16 path takes one summed and one concatenated input of 16-channel blocks, "
                  "no fused residual / reserve / carry");
    p.quads_per_slice = bf16_quads_per_slice(d->c0, d->c2, H);
    p.chq = 0;
    p.pad = 0;
    report(FNSSL_LSTM_FAMILY_BF16);
    return forward_bf16(p, H, st);
  }
  if (d->precision == FNSSL_PRECISION_BF16W) {   // wide bf16 kernels: 32 sequences per wave, bf16 / fp32 activation tensors
    FNSSL_REQUIRE(!(mode & (kHas1 | kSum)) && !d->reserve && !d->carry_state && d->c0 % 16 == 0 && d->c2 % 16 == 0,
                  "lstm_forward: the wide bf16 path takes 16-channel input blocks, no fused residual / reserve / carry");
    auto mult8 = [](long long v) { return (v & 7) == 0; };
    const int fm = d->f32_mask & 7;
    FNSSL_REQUIRE(((fm & 1) || !d->c0 || (mult8(d->src0.so) && mult8(d->src0.si) && mult8(d->src0.st))) &&
                      ((fm & 2) || !d->c2 || (mult8(d->src2.so) && mult8(d->src2.si) && mult8(d->src2.st))) &&
                      ((fm & 4) || (mult8(d->out_so) && mult8(d->out_si) && mult8(d->out_st))),
                  "lstm_forward: strides of bf16 tensors must be multiples of 8 elements");
    return forward_bf16w(p, H, fm, st, family);
  }
  FNSSL_REQUIRE(d->precision == FNSSL_PRECISION_FP32, "lstm_forward: unknown precision %d", d->precision);
  if (d->reserve) {   // training forward: also save the gate activations (lstm_train.hip)
    FNSSL_REQUIRE((H == 128 || H == 256) && !(mode & (kHas1 | kSum)),
                  "lstm_forward: the reserve-saving forward needs hidden 128/256 and no src1 / out_sum");
    FNSSL_REQUIRE(d->reserve_bytes >= fnssl_lstm_reserve_bytes(d->nseq, H, d->ndir, d->nsteps) &&
                      (reinterpret_cast<uintptr_t>(d->reserve) & 15) == 0 &&
                      (long double)d->nsteps * (H / 16) * kReserveRecs * 1024 < 4.0e9L,
                  "lstm_forward: reserve buffer too small or misaligned");
    p.chq = 0;
    p.pad = 0;
    p.task0 = 0;
    p.task1 = tasks;
    p.wgs_per_dir = 0;
    // the full-band layers of a large enough shard: the cluster-resident kernel with the reserve stores (lstm_f32c.h), its
    // guarded fallback = the split kernels below
    if (d->variant == 0 && f32c_handles(p, H, mode)) {
      const int rc = forward_f32c(p, H, mode, st);
      if (rc == FNSSL_OK) {
        report(FNSSL_LSTM_FAMILY_F32_CLUSTER);
        if (dry) return FNSSL_OK;
        guarded = true;
        p.guard = reinterpret_cast<const unsigned*>(p.cluster_ws);
      } else if (rc != kNoCluster) {
        return rc;
      }
    }
    report(FNSSL_LSTM_FAMILY_TRAIN);
    return dry ? FNSSL_OK : forward_save(p, H, mode, st);
  }

  // H = 128 full-band layers at full-chip size: hidden slices over clusters of 8 CUs, groups as work items (lstm_f32c.h)
  // — followed, in the same call, by the rounds below as its GUARDED fallback (they return at once unless the cluster
  // kernel recorded a hand-off it gave up on: include/fnssl.h, fnssl_lstm_forward)
  if (d->variant == 0 && !(mode & kHas1) && f32c_handles(p, H, mode)) {
    // streaming call: the cluster kernel advances the carried cell state in place — snapshot c_{-1} first, so that the guarded
    // fallback below can restart from it if the launch gives up (carry_backup_bytes)
    const size_t cell_bytes = cell_scratch_bytes(d->nseq, H, d->ndir);
    char* backup = p.cluster_ws + cluster_bytes(d->nseq, H, d->ndir, d->precision);
    const bool snap = p.carry && !dry && carry_backup_bytes(d->nseq, H, d->ndir, d->precision) >= cell_bytes;
    FNSSL_REQUIRE(!p.carry || dry || snap, "lstm_forward: no room for the carried cell state's snapshot (hidden %d)", H);
    if (snap) FNSSL_HIP(hipMemcpyAsync(backup, p.cscratch, cell_bytes, hipMemcpyDeviceToDevice, st));
    const int rc = forward_f32c(p, H, mode, st);
    if (rc == FNSSL_OK) {
      report(FNSSL_LSTM_FAMILY_F32_CLUSTER);
      if (rounds) *rounds = 1;
      if (dry) return FNSSL_OK;
      guarded = true;
      p.guard = reinterpret_cast<const unsigned*>(p.cluster_ws);
      if (snap) {
        const long long n4 = (long long)(cell_bytes / 16);
        hipLaunchKernelGGL(restore_cell_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, st, p.guard,
                           reinterpret_cast<const float4*>(backup), reinterpret_cast<float4*>(p.cscratch), n4);
        FNSSL_CHECK_LAUNCH("restore_cell_kernel");
      }
    } else if (rc != kNoCluster) {
      return rc;
    }
  }

  // one launch of `nw` waves per workgroup over the 16-sequence groups [t0, t1) of every direction
  auto launch_range = [&](int variant, int t0, int t1) -> int {
    const Variant& vr = kVariants[variant];
    p.task0 = t0;
    p.task1 = t1;
    p.wgs_per_dir = (t1 - t0 + vr.NW - 1) / vr.NW;
    const int nwg = p.wgs_per_dir * d->ndir;
    ++nlaunch;
    if (d->variant == 0 && !fnssl::tune(FNSSL_TUNE_LSTM_NO_STATIC)) {
      int rc = kNoStatic;
      if (H == 128) rc = launch_static_h128(p, mode, vr.NW, nwg, st);
      if (H == 256) {
        // the full-chip narrow-band rounds: the operand-ring kernel (lstm_static3.h: two hidden slices per pass, x_t and h_{t-1}
        // streamed, no register spills; block 1's 260-channel layer included) on a pair-interleaved copy of the stream (re-
        // ordered into the workspace first: 2 MB per direction, one tiny launch).  NO_STATIC3 keeps the one-slice kernel: A/B.
        // (Round 3's two-slice kernel with h_{t-1} in registers, lstm_static2.h, was removed in round 5: superseded.)
        const bool s3 = !p.ablate && !fnssl::tune(FNSSL_TUNE_NO_STATIC3) && (mode == kSum || mode == 0 || mode == (kHas2 | kSum)) &&
                        ((d->c2 == 0 && !(mode & kHas2)) || (d->c2 == 4 && (mode & kHas2)));
        if (vr.NW == 12 && d->c0 == 256 && !p.carry && s3) {
          LstmParams p2 = p;
          const long long n4 = (long long)(H / 16) * p.quads_per_slice * 4 * 64;      // float4 per direction
          char* dst = reinterpret_cast<char*>(d->workspace) + cell_scratch_bytes(d->nseq, H, d->ndir);
          // round 6: four slices per pass on a QUAD-interleaved copy (lstm_static4.h: half the operand re-reads, the weight ring
          // staged by LDS-DMA); NO_STATIC4 keeps the two-slice kernel: A/B, same bits
          const bool s4 = !fnssl::tune(FNSSL_TUNE_NO_STATIC4);
          for (int di = 0; di < d->ndir; ++di) {
            float4* o = reinterpret_cast<float4*>(dst + (size_t)di * n4 * 16);
            if (!dry) {
              if (s4)
                hipLaunchKernelGGL(quad_stream_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                                   reinterpret_cast<const float4*>(p.wpack[di]), p.quads_per_slice, n4, o);
              else
                hipLaunchKernelGGL(pair_stream_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                                   reinterpret_cast<const float4*>(p.wpack[di]), p.quads_per_slice, n4, o);
            }
            p2.wpack[di] = reinterpret_cast<const float*>(o);
          }
          FNSSL_CHECK_LAUNCH("pair_stream_kernel");
          rc = s4 ? launch_static4_h256(p2, mode, nwg, st) : launch_static3_h256(p2, mode, nwg, st);
          if (rc != kNoStatic) {
            report(FNSSL_LSTM_FAMILY_STATIC3);
            return rc;
          }
        }
        rc = launch_static_h256(p, mode, vr.NW, nwg, st);
      }
      if (rc == kNoStatic && !fnssl::tune(FNSSL_TUNE_NO_STATIC_IPDNET)) rc = launch_static_ipdnet(p, mode, H, vr.NW, nwg, st);
      if (rc != kNoStatic) {
        report(FNSSL_LSTM_FAMILY_STATIC);
        return rc;
      }
    }
    report(FNSSL_LSTM_FAMILY_GENERIC);
    p.chq = 0;
    p.pad = 0;
    if (vr.ring) choose_chunk(p.quads_per_slice, vr, p.chq, p.pad);
    switch (H) {
      case 16: return launch_h<16>(variant, p, mode, nwg, st);
      case 32: return launch_h<32>(variant, p, mode, nwg, st);
      case 64: return launch_h<64>(variant, p, mode, nwg, st);
      case 128: return launch_h<128>(variant, p, mode, nwg, st);
      default: return launch_h<256>(variant, p, mode, nwg, st);
    }
  };

  if (d->variant != 0 || H < 128) {
    const int variant = d->variant ? d->variant : default_variant(H);
    FNSSL_REQUIRE(variant >= 1 && variant <= kNumVariants, "lstm_forward: unknown variant %d", variant);
    const int rc = launch_range(variant, 0, tasks);
    if (rounds && !guarded) *rounds = nlaunch;
    return rc;
  }
  if (const int forced = default_variant_override(H)) {
    const int rc = launch_range(forced, 0, tasks);
    if (rounds && !guarded) *rounds = nlaunch;
    return rc;
  }

  const int ncu = fnssl::device_cus();
  // ---- few sequences (a single utterance, a streaming chunk): several waves per 16-sequence group ------
  auto launch_split = [&](int split, int t0, int t1, bool static_only) -> int {
    const int nw = split == 4 ? 8 : 4, groups_per_wg = nw / split;
    p.task0 = t0;
    p.task1 = t1;
    p.wgs_per_dir = (t1 - t0 + groups_per_wg - 1) / groups_per_wg;
    const Variant vr{nw, split == 4 ? 8 : 4, 1};
    choose_chunk(p.quads_per_slice, vr, p.chq, p.pad, split);
    const int nwg = p.wgs_per_dir * d->ndir;
    if (rounds) *rounds = 1;
    if (!p.carry && !(mode & kHas1)) {   // shape-specialised (ring-free) kernels for the network's own shapes
      const int rc = launch_split_static(p, H, nw, split, mode, 0, nwg, st);
      if (rc != kNoStatic) report(FNSSL_LSTM_FAMILY_SPLIT_STATIC);
      if (rc != kNoStatic || static_only) return rc;
    } else if (static_only) {
      return kNoStatic;
    }
    report(FNSSL_LSTM_FAMILY_SPLIT);
    return H == 128 ? launch_split_h<128>(split, p, mode, nwg, st) : launch_split_h<256>(split, p, mode, nwg, st);
  };
  {
    const long long total = (long long)tasks * d->ndir;
    if (const int f = fnssl::tune(FNSSL_TUNE_LSTM_SPLIT, 1, 4)) {
      if (f != 1) return launch_split(f == 3 ? 2 : f, 0, tasks, false);
    } else {
      // ring-free shape-specialised kernels (H = 256 layers of the network): 4 waves per group pay up to 6 groups
      // per CU (16 utterances: 13.9 -> 15.6 k frames/s); generic split kernels up to 2 groups per CU
      // (profiles/r01/d_batch_scan.txt)
      const int s4 = fnssl::tune(FNSSL_TUNE_SPLIT4_MAX_H256, 1, 64) ? fnssl::tune(FNSSL_TUNE_SPLIT4_MAX_H256, 1, 64) : 6;   // tuning knob
      if (H == 256 && total <= (long long)s4 * ncu) {
        const int rc = launch_split(4, 0, tasks, true);
        if (rc != kNoStatic) return rc;
      }
      if (total <= 2LL * ncu) return launch_split(4, 0, tasks, false);
    }
  }

  // ---- launch planner ---------------------------------------------------------------
  // Every wave does the same work, so a round (one launch, one workgroup per CU) runs at the pace of its fullest SIMD:
  // measured single rounds of the H = 128 layers (profiles/r02/n_h128_single_round_timings.txt) cost 36.8 ms at 12
  // waves per CU (3 per SIMD) but 46.6 / 46.7 / 47.5 / 48.9 ms at 13 / 14 / 15 / 16 (4 on the fullest SIMD).  The
  // per-CU wave count W = ceil(total groups / CUs) is therefore split into the CHEAPEST sequence of rounds by that
  // cost table (units: one wave-time per SIMD + a small per-wave term): config 2's full-band layers, 7200 groups ->
  // W = 29 -> 15 + 14 (no SIMD-balanced split covers 29); 191 of its 192 pairs, 7164 groups -> W = 28 -> 16 + 12, one
  // wave-time per SIMD less (Model.FN_SSL peels the last pair onto a second stream for exactly this reason).
  std::vector<int> seq_nw, seq_var;
  {
    const int rc = plan_lstm_rounds(H, tasks, d->ndir, ncu, seq_nw, seq_var);
    if (rc != FNSSL_OK) return rc;
  }
  const int wgs_per_dir_round = ncu / d->ndir > 0 ? ncu / d->ndir : 1;
  int t0 = 0;
  for (size_t r = 0; r < seq_nw.size() && t0 < tasks; ++r) {
    int t1 = r + 1 == seq_nw.size() ? tasks : t0 + wgs_per_dir_round * seq_nw[r];
    if (t1 > tasks) t1 = tasks;
    const int rc = launch_range(seq_var[r], t0, t1);
    if (rc != FNSSL_OK) return rc;
    t0 = t1;
  }
  if (rounds && !guarded) *rounds = nlaunch;
  return FNSSL_OK;
}

extern "C" {

int fnssl_lstm_forward(const fnssl_lstm_desc* d, void* stream) { return lstm_forward_impl(d, stream, false, nullptr, nullptr); }

int fnssl_lstm_plan(const fnssl_lstm_desc* d, int* family, int* rounds) {
  FNSSL_REQUIRE(family, "lstm_plan: null pointer");
  *family = 0;
  if (rounds) *rounds = 1;
  return lstm_forward_impl(d, nullptr, true, family, rounds);
}

}  // extern "C"

// ---- the generic recurrence kernels for hidden size 128 (explicit instantiation, see lstm_kernel.h; the other hidden sizes:
// lstm_hsmall.hip, lstm_h256.hip)
namespace fnssl_lstm {
template int launch_h<128>(int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
