// Host side of the LSTM training kernels (lstm_train.h): the launch planner shared by the
// reserve-saving forward and the BPTT kernel, the packer of the transposed weight stream, and the C ABI.
#include <cstring>

#include "common.h"
#include "tuning.h"

#include "lstm_train.h"
#include "lstm_split_static.h"
#include "lstm_bwdc.h"
#include "lstm_bwd2.h"
#include "lstm_fwd2.h"

using namespace fnssl_lstm;

namespace fnssl_lstm {

using fnssl::device_cus;

size_t bwdc_bytes(int nseq, int ndir);                                                    // lstm_bwdc.hip
bool bwdc_handles(const BwdParams& p, int H, BwdClusterParams& cp);
int backward_cluster(const BwdParams& p, BwdClusterParams cp, void* ws, hipStream_t st);

struct Geometry {
  int nw, split, t0, t1;
};

// Largest ring chunk (in super-quads) that lets every workgroup of a single-launch split geometry be
// resident at once: ceil(nwg / CUs) workgroups share a CU's 160 KB of LDS (2 ring slots each).
static int lds_chunk_cap(int nwg, int split) {
  const int ncu = device_cus();
  const int per_cu = (nwg + ncu - 1) / ncu;
  const int budget = (per_cu <= 1 ? 128 : per_cu >= 5 ? 32 : 160 / per_cu) * 1024;
  const int cap = budget / (2 * split * 4096);
  return cap > 0 ? cap : 1;
}

// Launch plan.  Plenty of 16-sequence groups: rounds of <= 12 waves per CU, as even as possible.  Fewer
// groups than would give every SIMD two waves: ONE launch in which 2 or 4 waves share a group (split), so
// the chip fills up with waves that each do a fraction of the per-step work.  FNSSL_TRAIN_SPLIT=1|2|4 forces.
template <class F>
static int plan_rounds(int tasks, int ndir, int max_split, int split4_groups_per_cu, F&& fn) {
  const int ncu = device_cus();
  const long long total = (long long)tasks * ndir;
  int split = 1;
  if (total <= (long long)split4_groups_per_cu * ncu)   // 2, or 6 for the ring-free H = 256 kernels
    split = 4;
  else if (total * 2 <= 12LL * ncu)
    split = 2;
  if (const int f = fnssl::tune(FNSSL_TUNE_TRAIN_SPLIT, 1, 4)) split = f == 3 ? 2 : f;
  if (split > max_split) split = max_split;
  if (split > 1) {
    // (one group per workgroup — 512 four-wave workgroups, two per CU, which drift apart — was measured at config 4:
    //  narrow-band BPTT 77.7 ms against 75.3 with two groups per workgroup)
    // (more groups than CUs: two groups per workgroup as well — 24 utterances per GPU = 384 narrow-band groups ran their
    //  BPTT as 384 one-group workgroups on 256 CUs in 72.4 ms, against 64.4 ms for config 4's 512 groups as 256 pairs)
    const int nw = (split == 4 && total > (long long)ncu) ? 8 : 4;
    return fn(Geometry{nw, split, 0, tasks});
  }
  const int W = (int)((total + ncu - 1) / ncu);
  const int rounds = (W + 11) / 12;
  const int wgs_per_dir_round = ncu / ndir > 0 ? ncu / ndir : 1;
  int t0 = 0;
  for (int r = 0; r < rounds && t0 < tasks; ++r) {
    const long long left_total = (long long)(tasks - t0) * ndir;
    const int want = (int)(((left_total + ncu - 1) / ncu + (rounds - r) - 1) / (rounds - r));
    const int nw = want <= 4 ? 4 : want <= 8 ? 8 : 12;
    int t1 = r + 1 == rounds ? tasks : t0 + wgs_per_dir_round * nw;
    if (t1 > tasks) t1 = tasks;
    const int rc = fn(Geometry{nw, 1, t0, t1});
    if (rc != FNSSL_OK) return rc;
    t0 = t1;
  }
  return FNSSL_OK;
}

// Training forward: called by fnssl_lstm_forward when the descriptor carries a reserve buffer.
int forward_save(LstmParams p, int H, int mode, hipStream_t st) {
  return plan_rounds(p.ntasks, p.ndir, 4, H == 256 ? 6 : 2, [&](const Geometry& gm) {
    p.task0 = gm.t0;
    p.task1 = gm.t1;
    const int groups_per_wg = gm.nw / gm.split;
    p.wgs_per_dir = (gm.t1 - gm.t0 + groups_per_wg - 1) / groups_per_wg;
    // H = 256 with four waves per group (config 4's narrow-band layers: two groups per CU): both groups of a CU against one
    // stream of weight records (lstm_fwd2.h)
    if (H == 256 && gm.split == 4 && gm.nw == 8 && p.c0 == 256 && !(mode & ~kHas2) && !fnssl::tune(FNSSL_TUNE_NO_FWD2)) {
      const bool has2 = (mode & kHas2) != 0;
      if ((!has2 && p.quads_per_slice == 33) || (has2 && p.c2 == 4 && p.quads_per_slice == 34)) {
        p.wgs_per_dir = (gm.t1 - gm.t0 + 1) / 2;
        const int nwg2 = p.wgs_per_dir * p.ndir;
        return has2 ? launch_fwd2_k<256, 16, 1>(p, nwg2, st) : launch_fwd2_k<256, 16, 0>(p, nwg2, st);
      }
    }
    const Variant vr{gm.nw, (gm.split > 1 && gm.nw == 8) ? 8 : 4, 1};   // staging registers as in launch_save_m
    const int nwg = p.wgs_per_dir * p.ndir;
    if (gm.split > 1 && !fnssl::tune(FNSSL_TUNE_TRAIN_NO_STATIC)) {   // shape-specialised kernels first
      const int rc = launch_split_static(p, H, gm.nw, gm.split, mode | kSave, lds_chunk_cap(nwg, gm.split), nwg, st);
      if (rc != kNoStatic) return rc;
    }
    choose_chunk(p.quads_per_slice, vr, p.chq, p.pad, gm.split, gm.split > 1 ? lds_chunk_cap(nwg, gm.split) : 0);
    return H == 128 ? launch_save<128>(gm.nw, gm.split, p, mode | kSave, nwg, st)
                    : launch_save<256>(gm.nw, gm.split, p, mode | kSave, nwg, st);
  });
}

}  // namespace fnssl_lstm

// carried dh / dc records (one region per direction and group, 16 spare), 256-byte aligned end
static size_t bwd_scratch_bytes(int nseq, int hidden, int ndir) {
  const size_t tasks = (size_t)(nseq + 15) / 16 + 16;
  return tasks * ndir * (size_t)(2 * (hidden / 16)) * 1024 + 256;
}
static int lstm_backward_impl(const fnssl_lstm_bwd_desc* d, void* stream, int dry, int* family);

extern "C" {

size_t fnssl_lstm_reserve_bytes(int nseq, int hidden, int ndir, int nsteps) {
  if (nseq <= 0 || hidden <= 0 || hidden % 16 || ndir <= 0 || nsteps <= 0) return 0;
  const size_t tasks = (size_t)(nseq + 15) / 16;
  return tasks * ndir * (size_t)nsteps * (hidden / 16) * kReserveRecs * 1024;
}

size_t fnssl_lstm_bwd_packed_floats(int c0g, int hidden) {
  if (hidden <= 0 || hidden % 16 || c0g < 0 || (c0g & 15)) return 0;
  return (size_t)(bwd_co_pad(c0g, hidden) / 64) * bwd_quads_per_slice(hidden) * 4 * 256;
}

int fnssl_lstm_pack_bwd(const float* w_ih, const float* w_hh, int c_in, int c0g, int H, float* packed) {
  FNSSL_REQUIRE(w_ih && w_hh && packed, "lstm_pack_bwd: null pointer");
  const size_t total = fnssl_lstm_bwd_packed_floats(c0g, H);
  FNSSL_REQUIRE(total > 0 && c0g <= c_in, "lstm_pack_bwd: unsupported sizes (c_in %d, c0g %d, H %d)", c_in, c0g, H);
  std::memset(packed, 0, total * sizeof(float));
  const int co = bwd_co_pad(c0g, H), hq = co / 4, nso = co / 64, nvb = 4 * H / 16;
  auto wb = [&](int o, int ku) -> float {   // [W_ih[:, :c0g] | W_hh]^T
    if (o < c0g) return w_ih[(size_t)ku * c_in + o];
    if (o < c0g + H) return w_hh[(size_t)ku * H + (o - c0g)];
    return 0.f;
  };
  float* rec = packed;
  for (int so = 0; so < nso; ++so) {
    rec += 4 * 256;   // "bias" quad: zeros (accumulator init)
    for (int v = 0; v < nvb; ++v)
      for (int j = 0; j < 4; ++j, rec += 256)
        for (int l = 0; l < 64; ++l)
          for (int qq = 0; qq < 4; ++qq)
            rec[l * 4 + qq] = wb(qq * hq + 16 * so + (l & 15), 16 * v + 4 * (l >> 4) + j);
  }
  if ((size_t)(rec - packed) != total) {
    fnssl::set_error("lstm_pack_bwd: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

size_t fnssl_lstm_bwd_workspace_bytes(int nseq, int hidden, int ndir) {
  if (nseq <= 0 || hidden <= 0 || ndir <= 0) return 0;
  return bwd_scratch_bytes(nseq, hidden, ndir) + bwdc_bytes(nseq, ndir);
}

int fnssl_lstm_backward(const fnssl_lstm_bwd_desc* d, void* stream) { return lstm_backward_impl(d, stream, 0, nullptr); }

int fnssl_lstm_backward_plan(const fnssl_lstm_bwd_desc* d, int* family) { return lstm_backward_impl(d, nullptr, 1, family); }

int fnssl_lstm_backward_status(const void* workspace, size_t workspace_bytes, int nseq, int hidden, int ndir, void* stream,
                               unsigned* status) {
  FNSSL_REQUIRE(workspace && status, "lstm_backward_status: null pointer");
  FNSSL_REQUIRE(workspace_bytes >= fnssl_lstm_bwd_workspace_bytes(nseq, hidden, ndir) && nseq > 0,
                "lstm_backward_status: not a workspace of this layer size");
  hipStream_t st = fnssl::as_stream(stream);
  FNSSL_HIP(hipMemcpyAsync(status, static_cast<const char*>(workspace) + bwd_scratch_bytes(nseq, hidden, ndir), sizeof(unsigned),
                           hipMemcpyDeviceToHost, st));
  FNSSL_HIP(hipStreamSynchronize(st));
  return FNSSL_OK;
}

}  // extern "C"

static int lstm_backward_impl(const fnssl_lstm_bwd_desc* d, void* stream, int dry, int* family) {
  FNSSL_REQUIRE(d, "lstm_backward: null descriptor");
  fnssl::TuningScope tuning_of_this_call(d->tuning);
  const int H = d->hidden;
  FNSSL_REQUIRE(H == 128 || H == 256, "lstm_backward: hidden size %d unsupported (128/256)", H);
  FNSSL_REQUIRE(d->ndir == 1 || d->ndir == 2, "lstm_backward: ndir must be 1 or 2");
  FNSSL_REQUIRE(d->nseq > 0 && d->nsteps > 0 && d->q_inner > 0, "lstm_backward: empty problem");
  FNSSL_REQUIRE(d->c0g >= 0 && d->c0g % 16 == 0, "lstm_backward: c0g %d must be a multiple of 16", d->c0g);
  FNSSL_REQUIRE(d->reserve && d->dh.p && d->da && d->wpack_bwd[0] && (d->ndir == 1 || d->wpack_bwd[1]) &&
                    (d->c0g == 0 || d->dx),
                "lstm_backward: null pointer");
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  auto mult4 = [](long long v) { return (v & 3) == 0 && v >= 0; };
  FNSSL_REQUIRE(aligned(d->reserve) && aligned(d->dh.p) && aligned(d->da) && aligned(d->dx) &&
                    aligned(d->wpack_bwd[0]) && aligned(d->wpack_bwd[1]) && aligned(d->workspace),
                "lstm_backward: pointers must be 16-byte aligned");
  FNSSL_REQUIRE(mult4(d->dh.so) && mult4(d->dh.si) && mult4(d->dh.st) && mult4(d->da_so) && mult4(d->da_si) &&
                    mult4(d->da_st) && mult4(d->dx_so) && mult4(d->dx_si) && mult4(d->dx_st),
                "lstm_backward: strides must be non-negative multiples of 4 floats");
  auto extent_ok = [&](long long so, long long si, long long st, long long width) {
    return ((long double)so + 16.0L * si + (long double)d->nsteps * st + width) * 4.0L < 4.0e9L;
  };
  FNSSL_REQUIRE(extent_ok(d->dh.so, d->dh.si, d->dh.st, 2 * H) && extent_ok(d->da_so, d->da_si, d->da_st, 8 * H) &&
                    (d->c0g == 0 || extent_ok(d->dx_so, d->dx_si, d->dx_st, 2 * d->c0g)) &&
                    (long double)d->nsteps * (H / 16) * kReserveRecs * 1024 < 4.0e9L,
                "lstm_backward: one sequence group must span < 4 GB");
  const size_t need = fnssl_lstm_bwd_workspace_bytes(d->nseq, H, d->ndir);
  if (!d->workspace || d->workspace_bytes < need) {
    fnssl::set_error("lstm_backward: workspace %zu < %zu bytes", d->workspace_bytes, need);
    return FNSSL_E_WORKSPACE;
  }
  BwdParams p;
  p.reserve = d->reserve;
  p.dh = View{d->dh.p, d->dh.so, d->dh.si, d->dh.st};
  p.da = d->da;
  p.da_so = d->da_so;
  p.da_si = d->da_si;
  p.da_st = d->da_st;
  p.dx = d->dx;
  p.dx_so = d->dx_so;
  p.dx_si = d->dx_si;
  p.dx_st = d->dx_st;
  p.wpack[0] = d->wpack_bwd[0];
  p.wpack[1] = d->wpack_bwd[1];
  p.scratch = static_cast<float*>(d->workspace);
  p.c0g = d->c0g;
  p.co_pad = bwd_co_pad(d->c0g, H);
  p.nseq = d->nseq;
  p.q_inner = d->q_inner;
  p.nsteps = d->nsteps;
  p.ndir = d->ndir;
  p.ntasks = (d->nseq + 15) / 16;
  p.quads_per_slice = bwd_quads_per_slice(H);
  p.dry = dry;
  p.fallback_count = d->fallback_count;
  hipStream_t st = dry ? nullptr : fnssl::as_stream(stream);
  const double flops = 2.0 * 4 * H * (double)(d->c0g + H) * d->nseq * (double)d->nsteps * d->ndir;
  fnssl::TimedLaunch tl(dry ? nullptr : H == 128 ? "lstm_bwd_h128" : "lstm_bwd_h256", st, flops);
  if (family) *family = FNSSL_LSTM_FAMILY_BWD;
  // full-band layers of a large enough shard: the cluster-resident kernel (lstm_bwdc.h), then — in the same call — the
  // kernels below as its guarded fallback (they return at once unless the cluster kernel recorded a hand-off it gave up on)
  BwdClusterParams cp{};
  if (bwdc_handles(p, H, cp)) {
    void* cws = static_cast<char*>(d->workspace) + bwd_scratch_bytes(d->nseq, H, d->ndir);
    const int rc = backward_cluster(p, cp, cws, st);
    if (rc == FNSSL_OK) {
      if (family) *family = FNSSL_LSTM_FAMILY_BWD_CLUSTER;
      if (dry) return FNSSL_OK;
      p.guard = static_cast<const unsigned*>(cws);
    } else if (rc != kNoCluster) {
      return rc;
    }
  }
  if (dry) return FNSSL_OK;
  const int nso = p.co_pad / 64;
  const int max_split = nso % 4 == 0 ? 4 : nso % 2 == 0 ? 2 : 1;   // output slices divide among the waves
  return plan_rounds(p.ntasks, p.ndir, max_split, H == 256 ? 6 : 2, [&](const Geometry& gm) {
    p.task0 = gm.t0;
    p.task1 = gm.t1;
    const int groups_per_wg = gm.nw / gm.split;
    p.wgs_per_dir = (gm.t1 - gm.t0 + groups_per_wg - 1) / groups_per_wg;
    // H = 256 with four waves per group (config 4's narrow-band layers: two groups per CU): both groups of a CU against one
    // stream of weight records (lstm_bwd2.h)
    if (H == 256 && gm.split == 4 && gm.nw == 8 && nso % 4 == 0 && !fnssl::tune(FNSSL_TUNE_NO_BWD2)) {
      p.wgs_per_dir = (gm.t1 - gm.t0 + 1) / 2;
      return launch_bwd2_k<256>(p, p.wgs_per_dir * p.ndir, st);
    }
    const Variant vr{gm.nw, gm.split > 1 ? 8 : 4, 1};
    const int nwg = p.wgs_per_dir * p.ndir;
    choose_chunk(p.quads_per_slice, vr, p.chq, p.pad, gm.split, gm.split > 1 ? lds_chunk_cap(nwg, gm.split) : 0);
    return H == 128 ? launch_bwd<128>(gm.nw, gm.split, p, nwg, st) : launch_bwd<256>(gm.nw, gm.split, p, nwg, st);
  });
}

// ---- the training kernels (forward with reserve, BPTT) for hidden size 128 (explicit instantiation, see lstm_train.h; hidden
// size 256: lstm_train_h256.hip)
namespace fnssl_lstm {
template int launch_bwd<128>(int, int, const BwdParams&, int, hipStream_t);
template int launch_save<128>(int, int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
