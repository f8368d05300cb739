// Host side of the cluster-resident BPTT kernel (lstm_bwdc.h): geometry, workspace carve-up, launch.
#include "common.h"
#include "tuning.h"

#include "lstm_bwdc.h"

namespace fnssl_lstm {

// bytes behind the carried-state records of fnssl_lstm_backward's workspace: status word (256 B) + tag arrays
size_t bwdc_bytes(int nseq, int ndir) {
  const size_t tasks = (size_t)(nseq + 15) / 16;
  return 256 + (tasks + 256) * (size_t)ndir * 16 * sizeof(unsigned);
}

// does the cluster kernel take this layer?  H = 128 and the output slices make whole clusters inside an XCD (config 4, block
// 1: 2 members, 9.4 groups per cluster — 10.7 against 13.4 ms on the split kernels).  Round 4 kept shards of fewer than 8
// groups per cluster on the split kernels; measured in round 5 (4 / 2 utterances per GPU: 4 / 2 groups per cluster) those
// take 31.1 / 30.5 ms for the three layers against 10.5 / 9.7 ms here — every group pulls the whole W^T from L2 per step —
// so the default threshold is 1 (knob BWD_CLUSTER_MIN_GROUPS).  BWD_NO_CLUSTER turns the kernel off.
bool bwdc_handles(const BwdParams& p, int H, BwdClusterParams& cp) {
  if (H != 128 || fnssl::tune(FNSSL_TUNE_BWD_NO_CLUSTER)) return false;
  // the kernel's addressing (lstm_bwdc.h locate()) takes a group's first sequence as its lowest address and lets a group
  // cross at most one outer index: the same preconditions f32c_handles() checks.  A permuted / time-major view or
  // q_inner < 16 goes to the split kernels, whose 64-bit address search takes any layout.
  auto grows = [&](long long so, long long si) { return si >= 0 && so >= (long long)(p.q_inner - 1) * si; };
  if (p.q_inner < 16 || !grows(p.dh.so, p.dh.si) || !grows(p.da_so, p.da_si) || (p.c0g != 0 && !grows(p.dx_so, p.dx_si)))
    return false;
  const int ncu = cluster_cus();   // device CUs minus fnssl_tuning RESERVED_CUS (RCCL's kernels under an overlapped backward)
  const int per_xcd = ncu / 8;
  cp.members = p.co_pad / 64;
  if (ncu % 8 || cp.members < 1 || cp.members > 8 || cp.members > per_xcd) return false;
  cp.clusters_per_xcd = per_xcd / cp.members;
  const int clusters = cp.clusters_per_xcd * 8;
  if (clusters % p.ndir) return false;
  cp.clusters_per_dir = clusters / p.ndir;
  cp.groups_per_cluster = (p.ntasks + cp.clusters_per_dir - 1) / cp.clusters_per_dir;
  const int min_groups = fnssl::tune(FNSSL_TUNE_BWD_CLUSTER_MIN_GROUPS, 1, 1 << 20);
  return cp.groups_per_cluster >= (min_groups ? min_groups : 1);
}

// ws = the region bwdc_bytes() sizes.  FNSSL_OK, kNoCluster, or an error.
int backward_cluster(const BwdParams& p, BwdClusterParams cp, void* ws, hipStream_t st) {
  cp.status = static_cast<unsigned*>(ws);
  cp.tags = cp.status + 64;
  cp.spin_limit = cluster_spin_limit();
  cp.stall_member = cluster_test_stall();
  cp.rotate = !fnssl::tune(FNSSL_TUNE_BWD_CLUSTER_NO_ROTATE);
  cp.no_prefetch = fnssl::tune(FNSSL_TUNE_BWDC_NO_PREFETCH);
  cp.simd_token = !fnssl::tune(FNSSL_TUNE_BWDC_NO_TOKEN);
  const size_t tag_bytes = (size_t)cp.clusters_per_dir * p.ndir * cp.groups_per_cluster * 16 * sizeof(unsigned);
  if (!p.dry) FNSSL_HIP(hipMemsetAsync(ws, 0, 256 + tag_bytes, st));   // (dry: the launchers stop after their occupancy check)
#ifdef FNSSL_BUILD_ABLATE
  if ((cp.ablate = env_int("FNSSL_BWDC_ABLATE", 1, 1 << 20)) != 0) return launch_bwdc_k<kBwdcWaves, true>(p, cp, st);
#endif
  if (fnssl::tune(FNSSL_TUNE_BWDC_WAVES16)) return launch_bwdc_k<16, false, 4>(p, cp, st);   // the first shape: 16 waves, 4-deep ring
  return launch_bwdc_k<kBwdcWaves>(p, cp, st);
}

}  // namespace fnssl_lstm
