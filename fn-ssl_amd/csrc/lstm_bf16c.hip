// Cluster-resident bf16 LSTM kernel (lstm_bf16c.h): instantiation for IPDnet's narrow-band layer shape and its launcher.
#include <algorithm>
#include <cstdlib>

#include "lstm_bf16c.h"
#include "tuning.h"

namespace fnssl_lstm {

// true when a cluster kernel is built for the shape (and not switched off: FNSSL_NO_CLUSTER=1 keeps the pair-split
// kernels of lstm_bf16p.h, which give the same bits — A/B, and the path of shapes / batches not covered here)
bool bf16c_handles(const LstmParams& p, int H, int flags) {
  if (fnssl::tune(FNSSL_TUNE_NO_CLUSTER)) return false;
  // Measured on MI355X (profiles/r03/j_*, "small batches"): the narrow-band kernel wins at every batch (12.5 against 19.7 us
  // per step from 512 sequences up: no weight stream at all); the full-band kernels walk three parts per step (~18 us)
  // where one round of pair-split workgroups takes ~10, so they take over when the pair-split launch would need more
  // than one workgroup per CU (config 3: 600 workgroups, 33.6 against 23.9 us per step).
  const long long pair_wgs = (long long)((p.nseq + 63) / 64) * p.ndir;
  const bool past_one_round = pair_wgs > fnssl::device_cus();
  if (H == 128 && p.c0 == 16 && p.c2 == 0 && flags == kW_F0)   // block 1's full-band layer: 16 fp32 feature channels
    return past_one_round && !fnssl::tune(FNSSL_TUNE_NO_CLUSTER_B1);
  if (p.c0 != 256 || p.c2 != 16 || flags != kW_F2) return false;
  if (H == 256) return true;
  if (H == 128) return past_one_round && !fnssl::tune(FNSSL_TUNE_NO_CLUSTER_H128);
  return false;
}

template <int H>
static int launch_one(const LstmParams& p, const ClusterParams& cp, hipStream_t st) {
  if constexpr (H == 128) {
    if (p.c0 == 0) return launch_bf16c_k<128, 0, 1, kW_F2>(p, cp, st);   // block 1 (its input travels as the fp32 block)
  }
#ifdef FNSSL_BUILD_ABLATE   // timing ablations (wrong results): make ABLATE=1 only
  switch (env_int("FNSSL_CLUSTER_ABL", 1, 127)) {
    case 1: return launch_bf16c_k<H, 16, 1, kW_F2, 1>(p, cp, st);
    case 2: return launch_bf16c_k<H, 16, 1, kW_F2, 2>(p, cp, st);
    case 9: return launch_bf16c_k<H, 16, 1, kW_F2, 9>(p, cp, st);
    case 25: return launch_bf16c_k<H, 16, 1, kW_F2, 25>(p, cp, st);
    case 32: return launch_bf16c_k<H, 16, 1, kW_F2, 32>(p, cp, st);
    case 65: return launch_bf16c_k<H, 16, 1, kW_F2, 65>(p, cp, st);
    default: break;
  }
#endif
  // (kernel template knobs WG_ / AD — a two-group operand window, A operands read two K-steps ahead — were measured at
  // config 3 and change nothing: profiles/r03/j_cluster_kernel_timing_and_ablations.txt; the defaults are built)
  // (a 64-sequences-per-wave formulation — one wave per SIMD, two accumulator sets, the previous part's gate math issued
  //  between the MFMAs — was built and measured: bit-identical, 8.3 ms against 5.7 for the narrow-band layer; the gate math
  //  of 64 sequences needs every issue cycle the MFMAs leave free, and one wave per SIMD exposes every store
  //  acknowledgement and tag wait: profiles/r03/l_cluster_64_sequences_per_wave_not_kept.txt, git history for the code)
  return launch_bf16c_k<H, 16, 1, kW_F2>(p, cp, st);
}

// Launches of at most CUs / members clusters (one workgroup per CU: every member of every cluster of a launch is resident).
int forward_bf16c(LstmParams p, int H, int flags, hipStream_t st) {
  if (flags == kW_F0) {   // block 1's full-band layer: the kernel's one fp32 block is src2; same record layout in the stream
    p.src2 = p.src0;
    p.c2 = p.c0;
    p.c0 = 0;
  }
  const int ncu = fnssl::device_cus();
  const int per_launch = ncu / cluster_members(H);
  FNSSL_REQUIRE(per_launch >= 1, "lstm_forward: the cluster kernel needs at least %d CUs", cluster_members(H));
  // How many clusters per direction.  A cluster holds at most 8 NP tiles of 32 sequences; tile 8 pt + w is part pt of wave w, and
  // waves w and w + 4 share a SIMD, so a step costs a SIMD parts(0) + parts(4) part-times (parts(w) = ceil((tpc - w) / 8), a
  // running wave cycles through at least two).  With three parts per wave (H = 128) the fullest clusters cost 3 + 3; cutting the
  // batch into MORE clusters of 17 - 20 tiles costs 3 + 2 while they still fit one launch (config 3's full-band layers: 1200
  // tiles, 50 clusters x 24 on 200 CUs -> 60 x 20 on 240).  The smallest count with the lowest cost is taken; 16 tiles or fewer
  // per cluster (2 + 2) would need a third more CUs than the batch's 17 tiles per cluster bound allows the workspace (lstm.hip).
  const int tiles = (p.nseq + 31) / 32, np = cluster_parts(H);
  int cl_per_dir = (tiles + 8 * np - 1) / (8 * np);
  auto cost = [&](int cl) {
    const int tpc = (tiles + cl - 1) / cl;
    auto parts = [&](int w) { const int n = tpc > w ? (tpc - w + 7) / 8 : 0; return n == 0 ? 0 : n < 2 ? 2 : n; };
    return parts(0) + parts(4);
  };
  if (np == 3 && !fnssl::tune(FNSSL_TUNE_CLUSTER_FULL_TILES)) {
    const int most = std::min(per_launch / p.ndir, tiles / 17);     // one launch, and >= 17 tiles per cluster
    int best = cl_per_dir;
    for (int cl = cl_per_dir + 1; cl <= most; ++cl)
      if (cost(cl) < cost(best)) best = cl;
    cl_per_dir = best;
  }
  const int ncl = cl_per_dir * p.ndir;
  const size_t head = 256 + (size_t)ncl * (kClusterTagWords * 4);
  ClusterParams cp;
  cp.status = reinterpret_cast<unsigned*>(p.cluster_ws);
  cp.tags = reinterpret_cast<unsigned*>(p.cluster_ws + 256);
  cp.hx = p.cluster_ws + head;
  FNSSL_REQUIRE((size_t)ncl * 2 * cluster_parity_bytes(H) < 0xf0000000ull, "lstm_forward: too many sequences for one call of the cluster kernel");
  cp.parity_stride = (unsigned)((size_t)ncl * cluster_parity_bytes(H));
  cp.cl_per_dir = cl_per_dir;
  cp.tpc = (tiles + cl_per_dir - 1) / cl_per_dir;
  // placement is a speed matter only; the knob puts the members of a cluster on DIFFERENT XCDs so that tests can show it
  cp.spread = fnssl::tune(FNSSL_TUNE_CLUSTER_SPREAD) ? 1 : 0;
  cp.spin_limit = cluster_spin_limit();
  cp.stall_member = cluster_test_stall();
  // status word + tags, and the parity-1 operand records (step 0 reads h_{-1} = 0 from them)
  if (!p.dry) {
    FNSSL_HIP(hipMemsetAsync(p.cluster_ws, 0, head, st));
    FNSSL_HIP(hipMemsetAsync(cp.hx + cp.parity_stride, 0, cp.parity_stride, st));
  }
  for (int c0 = 0; c0 < ncl; c0 += per_launch) {
    cp.cl0 = c0;
    cp.ncl = ncl - c0 < per_launch ? ncl - c0 : per_launch;
    const int rc = H == 256 ? launch_one<256>(p, cp, st) : launch_one<128>(p, cp, st);
    if (rc != FNSSL_OK) return rc;
  }
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
